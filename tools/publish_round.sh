#!/bin/bash
# Turns the outputs of the final GPU calls (merged into gpurun_out/) into the round's files under profiles/.  Run in the container,
# from the repo root, after:
#   gpurun -- 'bash tools/gpu_round.sh tests smoke fuzz:300:9601:pre fuzz:300:9701:gen fuzz:100:9801:hmm'
#   gpurun -- 'bash tools/gpu_round.sh profiles:r06 quantprof "profiles:r06np:--no-prefilter" "profiles:r06m512:--codebook-size 512" \
#              "profiles:r06m256:--codebook-size 256" stamps exp:0,1,2,3,0'
#   (commit the summaries, then)  gpurun -- 'bash tools/gpu_round.sh bench'     -- so that the line quotes the fresh traffic file
# usage: tools/publish_round.sh [summaries|bench|all]
set -e
cd "$(dirname "$0")/.."
WHAT=${1:-all}
TAG=${TAG:-r06}
H=$(python -c "import bench; print(bench.kernel_sources_sha16())")
if [ "$WHAT" = summaries ] || [ "$WHAT" = all ]; then
  for t in $TAG ${TAG}np ${TAG}m512 ${TAG}m256; do
    python tools/summarize_profiles.py $t gpurun_out profiles > /tmp/sum_$t.log 2>&1 || { echo "summarize $t failed"; tail -5 /tmp/sum_$t.log; exit 1; }
  done
  python tools/summarize_quantize_profile.py ${TAG}q > /tmp/sum_${TAG}q.log 2>&1 || { tail -5 /tmp/sum_${TAG}q.log; exit 1; }
  python - "$H" "$TAG" <<'PY'
import re, sys
H, TAG = sys.argv[1], sys.argv[2]
p = f"profiles/{TAG}_sweep_experiments.txt"
s = open(p).read()
s = re.sub(r"final sources \([0-9a-f]{16}\)", f"final sources ({H})", s)
a, b = s.index("M   256 exp  0"), s.index("Reading it (M = 1024):")
s = s[:a] + "\n".join(l[:330] for l in open(f"gpurun_out/{TAG}_exp.txt").read().strip().split("\n")) + "\n\n" + s[b:]
a = s.index("\n", s.index("== phase stamps of the final kernel")) + 1
s = s[:a] + "\n".join(l[:420] for l in open(f"gpurun_out/{TAG}_stamps.txt").read().split("\n") if "pass 1" not in l).strip() + "\n"
open(p, "w").write(s)
f = open(f"profiles/{TAG}_fuzz.txt").read()
f = re.sub(r"final sources \([0-9a-f]{16}\)", f"final sources ({H})", f)
f = re.sub(r"\d+ passed, \d+ skipped, \d+ deselected in [^\n]*", open(f"gpurun_out/{TAG}_tests.log").read().strip().split("\n")[-1], f)
for mode, head in (("pre", "prefilter fuzz done"), ("gen", "fuzz done"), ("hmm", "hmm fuzz done")):
    last = open(f"gpurun_out/{TAG}_fuzz_{mode}.txt").read().strip().split("\n")[-1]
    f = re.sub(r"^" + head + r"[^\n]*$", last, f, count=1, flags=re.M)  # (the first call's line; the "more of it" section is edited by hand)
open(f"profiles/{TAG}_fuzz.txt", "w").write(f)
for p in ("profiles/README.md",):
    x = open(p).read()
    x = re.sub(r"Round 6 \(final build, sources `[0-9a-f]{16}`", f"Round 6 (final build, sources `{H}`", x)
    open(p, "w").write(x)
PY
  python - <<'PY'
import json
for t, sfx in (("r06", ""), ("r06np", ""), ("r06m512", "_M512"), ("r06m256", "_M256")):
    d = json.load(open(f"profiles/{t}_pass_kernel{sfx}.json"))
    tn = {"r06np": "_noprefilter"}.get(t, sfx)
    tr = json.load(open(f"profiles/{t}_traffic{tn}.json")); sq = json.load(open(f"profiles/{t}_sq_counters{sfx}.json"))
    print(t, "avg %.4f" % d["avg_ms"], [round(x, 4) for x in d["avg_ms_by_pass_of_level"]], "events %.4f" % d["bench_kernel_ms"], "value %.3f G" % (d["bench_value"] / 1e9),
          "| fetch %.3f write %.3f GB ratio %.2f | pipe busy %.3f clock %.2f" % (tr["fetch_bytes"] / 1e9, tr["write_bytes"] / 1e9, tr["hbm_bytes_per_launch"] / tr["algorithmic_bytes_per_launch"], sq["mfma_pipe_busy_fraction"], sq["clock_GHz_under_pmc"]), d["kernel_sources_sha16"])
q = json.load(open("profiles/r06q_pass_kernel.json")); qs = json.load(open("profiles/r06q_sq_counters.json")); qt = json.load(open("profiles/r06q_traffic.json"))
print("r06q avg %.4f events %.4f %.3f G frac %.3f traffic x%.3f pipe busy %.3f" % (q["avg_ms"], q["event_ms_of_the_same_calls"], q["frames_per_sec_by_events"] / 1e9, q["frac_of_2500_TF"], qt["ratio_to_algorithmic"], qs["mfma_pipe_busy_fraction"]), q["kernel_sources_sha16"])
PY
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  python - "$H" "$TAG" <<'PY'
import json, subprocess, sys
H, TAG = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open(f"gpurun_out/{TAG}_bench.json") if l.startswith("{")][-1])
json.dump(d, open(f"profiles/{TAG}_bench.json", "w"), indent=1)
print("roofline.traffic:", d["roofline"]["traffic"], "stale:", (d["roofline"].get("traffic_detail") or {}).get("stale"))
dig = subprocess.run(["python", "tools/bench_digest.py", f"python bench.py   (default: N = 1, BASELINE config 4 whole on one GPU)=gpurun_out/{TAG}_bench.json"],
                     capture_output=True, text=True).stdout
open(f"profiles/{TAG}_bench_digest.txt", "w").write(f"round 6, final sources ({H}): the default bench line of one MI355X box, digest by tools/bench_digest.py "
                                                     f"(the line itself: profiles/{TAG}_bench.json)\n\n" + dig)
PY
fi
echo "published for sources $H"
