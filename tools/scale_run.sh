#!/bin/bash
# Weak-scaling sweep of bench.py on ONE node: N = 1, 2, 4, 8 ranks (one process per GPU, RCCL over
# xGMI), 32 clips per GPU, moco-k 16384 at N > 1 (BASELINE.json configs[2]; the N=1 point on that queue
# size is `value_k16384` of the N=1 line), then the shuffle-BN exchange variants at the largest N.  The
# builder had one GPU: this script is what a node with 8 of them should run; every line it prints is
# bench.py's own JSON line (value = whole-job clips/s) and, under it, THE LADDER: which rung the run ended
# on (bench_multi.RUNG_NAMES: 0 = everything on), what the self-check found on each rung, which exchange
# scheme COCLR_SHUFFLE=auto chose, and every attempt the supervisors made.
#   tools/scale_run.sh [max_gpus=8] [steps=20] [warmup=5]
set -u
cd "$(dirname "$0")/.."
MAXN=${1:-8}; STEPS=${2:-20}; WARM=${3:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${SCALE_OUT:-gpurun_out/scale}
mkdir -p "$OUT"
run() {   # run <n> <tag> [env...]
  local n=$1 tag=$2; shift 2
  local port=$((29600 + RANDOM % 300))
  if [ "$n" = 1 ]; then
    env "$@" python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-extra-legs \
      > "$OUT/$tag.log" 2> "$OUT/$tag.err"
  else
    env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 \
      --master-port "$port" bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARM" --no-extra-legs \
      > "$OUT/$tag.log" 2> "$OUT/$tag.err"
  fi
  echo "== $tag (rc=$?)"; tail -1 "$OUT/$tag.log" | cut -c1-400
}
for n in 1 2 4 8; do
  [ "$n" -le "$MAXN" ] || break
  run "$n" "n${n}_auto" COCLR_QUIET=1                    # the default: pull if verified, else routed
done
N=$MAXN
if [ "$N" -gt 1 ]; then
  run "$N" "n${N}_routed" COCLR_SHUFFLE=routed           # RCCL all_to_all_single of exactly the clips needed
  run "$N" "n${N}_allgather" COCLR_SHUFFLE=allgather     # the reference's own exchange
  run "$N" "n${N}_nohook" COCLR_DDP_HOOK=0               # DDP's per-parameter bucket copies
fi
python - "$OUT" <<'PY'
import json, sys, glob, os
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.log"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows[os.path.basename(f)[:-4]] = d
base = rows.get("n1_auto")
k16 = ((base or {}).get("value_k16384") or {}).get("value") or (base or {}).get("value")
for tag, d in rows.items():
    n, v, ms = d["n_gpus"], d["value"], d["ms_per_step"]
    eff = "" if not (k16 and v) else "  efficiency vs the N=1 K=16384 point: %.2f" % (v / (k16 * n))
    print("%-14s n=%d  %9s clips/s  %7s ms/step%s" % (tag, n, v, ms, eff))
    mg, sc = d.get("multi_gpu") or {}, d.get("self_check") or {}
    if mg:
        print("    rung %s (%s); exchange %s (%s); attempts %s" % (
            mg.get("rung"), mg.get("rung_what"), mg.get("shuffle_mode"),
            (mg.get("shuffle_selection") or {}).get("why"),
            [(a["started_on_rung"], a["ok"]) for a in mg.get("attempts", [])]))
    if sc:
        print("    self-check: %s" % [(t["rung"], t["bit_identical_to_serial_on_every_rank"]) for t in sc["trials"]])
PY
