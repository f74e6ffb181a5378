"""Markdown table of the psd:: kernels in a rocprofv3 `--kernel-trace --stats --output-format csv` run.
usage: python tools/kernel_stats_md.py <dir>/<prefix>_kernel_stats.csv "title" """
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("psd::", "void psd::")) or "rocclr" in r["Name"]]
print(f"# {sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]}\n\n| kernel | calls | total ms | avg us | min us | max us |\n|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} |")
