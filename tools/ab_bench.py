"""A/B kernel variants under raytracinginoneweekend_amd/variants/ in one process each, interleaved rounds.
usage: python tools/ab_bench.py <spp> <rounds> <variant> [<variant> ...]      (run on the GPU box)
Variants are named explicitly: only builds that already passed the GPU parity tests (tools/gpu_session.sh variants)
may be timed — a build that was never checked is never picked up by a glob."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 4:
    raise SystemExit(__doc__)
spp = sys.argv[1]
rounds = int(sys.argv[2])
libs = [os.path.join(ROOT, "raytracinginoneweekend_amd", "variants", "librtk_%s.so" % n) for n in sys.argv[3:]]
for lib in libs:
    if not os.path.exists(lib):
        raise SystemExit("missing " + lib)
res = {}
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, RTK_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--spp", spp, "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            # a variant that crashed may have faulted the GPU: start nothing more on it in this call
            print(os.path.basename(lib), "FAILED (exit %d); stopping here" % out.returncode, out.stderr[-600:])
            sys.exit(1)
        j = json.loads(line[-1])
        res.setdefault(os.path.basename(lib), []).append((j["value"], j["roofline"]["frac"], j["roofline"]["exact_tests_per_ray"], j["roofline"]["lane_utilisation"], j["roofline"]["kernel_ms"]))
        print(os.path.basename(lib), "round", r, "%.0f Mrays/s, kernel %.2f ms" % (j["value"], j["roofline"]["kernel_ms"]), flush=True)
for k, v in res.items():
    print("%-28s Mrays/s %s  frac %s  kernel ms %s  exact/ray %.2f util %.3f" % (k, ["%.0f" % x[0] for x in v], ["%.3f" % x[1] for x in v], ["%.2f" % x[4] for x in v], v[0][2], v[0][3]))
