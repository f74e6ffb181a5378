"""Poll socket power and shader clock (amdgpu hwmon sysfs, falling back to rocm-smi) while a command runs.
usage: python tools/power_poll.py <label> -- <command...>      prints one summary line (JSON)"""
import glob, json, subprocess, sys, threading, time

label = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]


def find(pattern):
    out = []
    for p in glob.glob(pattern):
        try:
            open(p).read()
            out.append(p)
        except OSError:
            pass
    return out


pw = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
fq = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
cap = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")
samples = []
stop = False


def poll():
    while not stop:
        row = {}
        try:
            if pw:
                row["w"] = int(open(pw[0]).read()) / 1e6
            if fq:
                row["mhz"] = int(open(fq[0]).read()) / 1e6
        except (OSError, ValueError):
            pass
        if row:
            samples.append(row)
        time.sleep(0.02)


t = threading.Thread(target=poll, daemon=True)
t.start()
t0 = time.time()
rc = subprocess.call(cmd)
dt = time.time() - t0
stop = True
t.join()
ws = sorted(s["w"] for s in samples if "w" in s)
fs = sorted(s["mhz"] for s in samples if "mhz" in s)
out = {"label": label, "rc": rc, "seconds": round(dt, 2), "n": len(samples), "sysfs": bool(pw)}
# the command spends its first seconds setting up; the steady state is the second half of the samples
half = samples[len(samples) // 2:]
ws2 = sorted(s["w"] for s in half if "w" in s)
fs2 = sorted(s["mhz"] for s in half if "mhz" in s)
if ws:
    out.update(power_w_max=ws[-1], power_w_median_2nd_half=ws2[len(ws2) // 2], power_w_min_2nd_half=ws2[0])
if fs:
    out.update(sclk_mhz_median_2nd_half=fs2[len(fs2) // 2], sclk_mhz_min_2nd_half=fs2[0], sclk_mhz_max=fs[-1])
out["series_w"] = [round(s.get("w", 0)) for s in samples[::10]]
out["series_mhz"] = [round(s.get("mhz", 0)) for s in samples[::10]]
if cap:
    try:
        out["power_cap_w"] = int(open(cap[0]).read()) / 1e6
    except (OSError, ValueError):
        pass
print(json.dumps(out))
