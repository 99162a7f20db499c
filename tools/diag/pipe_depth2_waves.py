"""Diagnostic: bench.py pipelined at depth 2 with other wave shapes of the side-stream pack kernel."""
import os, sys, subprocess, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for w in (4, 2, 8, 1):
    code = f"import sys; sys.path.insert(0, {root!r}); import cave_amd.qpsolver as q; q.PIPE_PACK_WAVES = {w}; sys.argv = ['bench.py', '--pipeline', '--pipeline-depth', '2', '--no-extras', '--no-other-configs', '--cpu-sample', '0', '--steps', '200']; import runpy; runpy.run_path({root!r} + '/bench.py', run_name='__main__')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print("pack waves", w, round(json.loads(line[-1])["ms_per_step"] * 1e3, 1) if line else r.stderr[-300:], flush=True)
