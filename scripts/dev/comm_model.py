"""Dev tool: bench.py's comm_model leg alone."""
import json, sys
sys.path.insert(0, ".")
import bench
class A: ni, nj, nk, dt = 1440, 1080, 75, 900.0
print(json.dumps(bench.comm_model_leg(A(), 0), indent=1))
