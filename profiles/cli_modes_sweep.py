"""The `dipper` command's own runtime settings (HSA_ENABLE_SDMA=0, GPU_MAX_HW_QUEUES=2, main.cpp) against the runtime's defaults
(DPR_CLI_RUNTIME_DEFAULTS=1) in every mode: wall time of the whole command, interleaved runs.  python profiles/cli_modes_sweep.py [runs]"""
import json, os, statistics, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tmp = tempfile.mkdtemp(prefix="climo_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
gen = os.path.join(ROOT, "tools", "bin", "gen_synth")
exe = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
def make(tag, tips, sites, extra):
    fa = os.path.join(tmp, tag + ".fa")
    subprocess.run([gen, "--tips", str(tips), "--sites", str(sites), "--seed", "3", "--mean-bl", "1e-3", "--lo", "1e-4", "--hi", "1e-2", "--fasta", fa] + extra, check=True)
    return fa
cases = [("NJ 30000 x 10000 (-m 2)", make("nj", 30000, 10000, []), ["-i", "m", "-m", "2"]),
         ("placement 60000 aligned x 2000 (-m 1)", make("pl", 60000, 2000, []), ["-i", "m", "-m", "1"]),
         ("placement 60000 unaligned x ~3000 (-i r -m 1)", make("rd", 60000, 3000, ["--indel", "0.03,0.09"]), ["-i", "r", "-m", "1"]),
         ("divide-and-conquer 400000 aligned x 400 (-m 3)", make("dc", 400000, 400, []), ["-i", "m", "-m", "3"])]
for name, fa, args in cases:
    res = {"tuned": [], "defaults": []}
    for r in range(runs + 1):
        for variant in ("tuned", "defaults"):
            env = dict(os.environ)
            if variant == "defaults":
                env["DPR_CLI_RUNTIME_DEFAULTS"] = "1"
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-I", fa, "-O", os.path.join(tmp, "o.nwk"), "-d", "2"] + args, capture_output=True, text=True, env=env)
            wall = (time.perf_counter() - t0) * 1e3
            if p.returncode != 0:
                print(name, variant, "FAILED", p.stderr[-300:], flush=True)
                break
            if r > 0:
                res[variant].append(round(wall))
    print(json.dumps({"case": name, "tuned_ms_median": statistics.median(res["tuned"]) if res["tuned"] else None,
                      "defaults_ms_median": statistics.median(res["defaults"]) if res["defaults"] else None, **res}), flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
