"""Diagnostic: bench.py pipelined at depth 1 / 2 with other spacer lengths."""
import os, sys, subprocess, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for depth, sp in ((1, 25000), (2, 25000), (2, 0), (2, 10000), (1, 15000), (1, 40000)):
    code = f"import sys; sys.path.insert(0, {root!r}); import cave_amd.qpsolver as q; q.PIPE_SPACER_CYCLES = {sp}; sys.argv = ['bench.py', '--pipeline-depth', '{depth}', '--no-extras', '--no-other-configs', '--cpu-sample', '0', '--steps', '200']; import runpy; runpy.run_path({root!r} + '/bench.py', run_name='__main__')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1]); print("depth", depth, "spacer", sp, "ms_per_step", round(j["ms_per_step"] * 1e3, 1), "us", flush=True)
    else:
        print("depth", depth, "spacer", sp, "failed", r.stderr[-400:], flush=True)
