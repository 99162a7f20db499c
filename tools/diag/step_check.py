"""Fused step kernel (cave_hip_cone_step): parity with the general operator and step times of its forms.

    python tools/diag/step_check.py [--tsp 20] [--batch 1024] [--steps 200] [--rotate 4] [--form fused|b2b|solve|pack|all]

forms:  fused  one launch per step = solve of batch i + pack of batch i+1 (the bench default)
        b2b    two launches per step on one stream: pack-only, then solve-only
        solve  the solve-only launch alone (cones already packed)
        pack   the pack-only launch alone
(replaces the round-3 one-offs pipe_*.py / cu_mask_*.py / overlap_probe.py, which probed the side-stream pipeline)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from cave_amd import _lib, synth
if os.environ.get("CAVE_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_LIB"])  # A/B builds of the library
from cave_amd import qpsolver
from cave_amd.qpsolver import PreparedCones, cone_op_dense, cone_op_prepared, prepare_dense

ap = argparse.ArgumentParser()
ap.add_argument("--tsp", type=int, default=20)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--rotate", type=int, default=4)
ap.add_argument("--form", default="all")
a = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.load()
R, B = a.rotate, a.batch
ctrs_np, costs_np, _ = synth.tsp_batch(a.tsp, R * B, seed=0)
rng = np.random.default_rng(1234)
batches = []
for r in range(R):
    ids = np.arange(B) + r * B
    pred = costs_np[ids] + rng.normal(0, 0.05, size=costs_np[ids].shape).astype(np.float32)
    batches.append((torch.tensor(ctrs_np[ids], device=dev), torch.tensor(pred, device=dev)))
outs = ("proj", "rnorm", "target", "loss", "grad")
mode = _lib.MODE_INNER
print("step lds bytes", qpsolver.step_lds_bytes(*batches[0][0].shape[1:]))
# ---- parity: fused chain vs the general operator
worst = 0.0
prep = prepare_dense(batches[0][0])
assert isinstance(prep, PreparedCones), "shape does not qualify"
for i in range(2 * R):
    c, p = batches[i % R]
    prep.then(batches[(i + 1) % R][0])
    got = cone_op_prepared(prep, p, mode, -1.0, 0.2, outputs=outs)
    ref = cone_op_dense(c, p, mode, -1.0, 0.2, outputs=outs, waves=2)
    for k in outs:
        worst = max(worst, float((got[k] - ref[k]).abs().max()))
    assert bool((got["status"] == 0).all()), got["status"].unique()
    prep = prep.next
    assert isinstance(prep, PreparedCones)
print(f"parity vs cone_op_dense (2 waves, general solver): max |diff| {worst:.3e}; iters mean {got['iters'].float().mean():.2f} max {int(got['iters'].max())}")

def timed(fn, n):
    for i in range(10): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t) / n

lo = ("loss", "grad")
st = {"prep": None}
def fused(i):
    if st["prep"] is None: st["prep"] = prepare_dense(batches[i % R][0])
    pr = st["prep"].then(batches[(i + 1) % R][0])
    cone_op_prepared(pr, batches[i % R][1], mode, -1.0, 0.2, check=False, outputs=lo)
    st["prep"] = pr.next
def b2b(i):
    pr = prepare_dense(batches[i % R][0])
    cone_op_prepared(pr, batches[i % R][1], mode, -1.0, 0.2, check=False, outputs=lo)
held = [prepare_dense(batches[r][0]) for r in range(min(R, 2))]
def solve(i): cone_op_prepared(held[i % len(held)], batches[i % len(held)][1], mode, -1.0, 0.2, check=False, outputs=lo)
def pack(i): prepare_dense(batches[i % R][0])
def old(i): cone_op_dense(batches[i % R][0], batches[i % R][1], mode, -1.0, 0.2, check=False, outputs=lo)
forms = {"fused": fused, "b2b": b2b, "solve": solve, "pack": pack, "general_split": old}
for name, fn in forms.items():
    if a.form in ("all", name):
        if name == "solve": held = [prepare_dense(batches[r][0]) for r in range(min(R, 2))]
        print(f"{name:14s} {timed(fn, a.steps):8.1f} us per step")
