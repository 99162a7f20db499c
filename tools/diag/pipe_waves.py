"""Diagnostic: bench.py --pipeline with other wave shapes of the side-stream pack kernel."""
import os, sys, subprocess, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for w in (4, 2, 1, 8):
    code = f"import sys; sys.path.insert(0, {root!r}); import cave_amd.qpsolver as q; q.PIPE_PACK_WAVES = {w}; q.SPLIT_NNZ = int(__import__('os').environ.get('SPLIT_NNZ', q.SPLIT_NNZ)); sys.argv = ['bench.py', '--pipeline', '--no-extras', '--no-other-configs', '--cpu-sample', '0', '--steps', '200']; import runpy; runpy.run_path({root!r} + '/bench.py', run_name='__main__')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1]); print("pack waves", w, "ms_per_step", round(j["ms_per_step"] * 1e3, 1), "us", flush=True)
    else:
        print("pack waves", w, "failed", r.stderr[-400:], flush=True)
