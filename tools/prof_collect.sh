#!/bin/bash
# Run ON THE GPU BOX (through gpurun): one rocprofv3 pass per counter group over the same bench.py command, CSV output
# under gpurun_out/prof_<tag>/.  Counter passes carry --kernel-trace only (no other trace domain), one group per run
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots, FETCH_SIZE and WRITE_SIZE do not fit one pass).
# tools/pmc_summary.py turns the directories into the tracked summaries under profiles/.
#   usage: tools/prof_collect.sh <tag> [bench.py arguments ...]
set -o pipefail
tag=${1:?tag}
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$tag
mkdir -p "$OUT"
python3 "$ROOT/tools/source_hash.py" > "$OUT.source_sha256" 2>/dev/null   # which build this output describes (tools/provenance.py)
cd /tmp && export TMPDIR=/tmp
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then ARGS=(--no-cpu-baseline --no-host-path --no-live-traffic --steps 6 --warmup 2 --pipeline 1); fi

run() {   # name, rocprofv3 options...
    local name=$1
    shift
    echo "[prof_collect] pass $name: $*"
    rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o run -- python3 "$ROOT/bench.py" "${ARGS[@]}" > "$OUT/$name.log" 2>&1 || {
        echo "[prof_collect] pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
    tail -1 "$OUT/$name.log" | cut -c1-200
}

# PROF_PASSES="stats sq1 sq2" restricts the passes (default: all six)
PASSES=${PROF_PASSES:-stats sq1 sq2 fetch write tcc}
rc=0
for pass in $PASSES; do
    case $pass in
        stats) run stats --kernel-trace --stats ;;
        sq1) run sq1 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE ;;
        sq2) run sq2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD ;;
        fetch) run fetch --kernel-trace --pmc FETCH_SIZE ;;
        write) run write --kernel-trace --pmc WRITE_SIZE ;;
        tcc) run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ;;
        *) echo "[prof_collect] unknown pass $pass"; false ;;
    esac || { rc=1; break; }
done
# keep only what the summary needs (the merge back is capped at 64 MiB)
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT" | cut -f1
exit $rc
