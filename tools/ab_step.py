#!/usr/bin/env python3
"""A/B of environment switches on the G+D step, arms alternating in fresh child processes (the switches are read once per
process):  python tools/ab_step.py [--rounds 3] [--steps 30] "PDGN_SPLIT_D=0" "PDGN_SPLIT_D=1" "PDGN_SPLIT_D=1,PDGN_G1_AHEAD=1" """
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, steps = 3, "30"
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    elif args[0] == "--steps":
        steps = args[1]
    args = args[2:]
arms = args
res = {a: [] for a in arms}
for r in range(rounds):
    for arm in arms:
        env = dict(os.environ)
        for kv in arm.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "5", "--no-cpu-baseline",
                            "--no-roofline", "--no-eval-c5"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(p.stderr[-2000:])
            sys.exit(1)
        res[arm].append(json.loads(line[0])["ms_per_step"])
        print("%-40s %.2f ms/step" % (arm or "(default)", res[arm][-1]), flush=True)
for arm in arms:
    v = sorted(res[arm])
    print("%-40s min %.2f  median %.2f  (%s)" % (arm or "(default)", v[0], v[len(v) // 2], " ".join("%.2f" % x for x in res[arm])))
