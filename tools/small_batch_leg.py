"""bench.py's small_batch_sampling record on its own (batch-1 / batch-5 DDPM steps + the reference's two whole calls)."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
sys.argv = ["bench.py"]
spec.loader.exec_module(b)
print(json.dumps(b.small_batch_leg(None)))
