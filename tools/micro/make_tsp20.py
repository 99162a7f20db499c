import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cave_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c, y, _ = synth.tsp_batch(n, 256, seed=0)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tsp20.bin"), "wb") as f:
    np.array(c.shape, np.int32).tofile(f); c.tofile(f); y.tofile(f)
