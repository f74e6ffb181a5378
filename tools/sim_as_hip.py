"""pytest plugin: the `hip_engine` of the GPU tests becomes the host-memory stand-in of the device engine (fuzz_host_vs_reference.sim_engine), so
that the GPU tests that only use the SceneManager / detector level run on CPU:  PYTHONPATH=tools python -m pytest -p sim_as_hip tests -m gpu -k ...
(tests that call the engine's own entry points fail over it with AttributeError: it has none).  Test infrastructure."""
import sys
_ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
sys.path[:0] = [__import__('os').path.join(_ROOT, 'tools'), __import__('os').path.join(_ROOT, 'oracle', 'cv2_shim'), _ROOT]
def pytest_configure(config):
    import fuzz_host_vs_reference as F
    from oracle.detectors_np import OracleEngine
    import pyscenedetect_amd.engine as E
    real = E.ScoringEngine
    def factory(device=0):
        sim = F.sim_engine(OracleEngine())
        sim.close = lambda: None
        return sim
    E.ScoringEngine = factory
