#!/bin/bash
# Regenerate EVERY profiles/rNN_* summary from the current binary on the GPU box (one command):
#   gpurun -- 'bash tools/make_profiles.sh r03'
# writes gpurun_out/profiles_<tag>/; copy that directory's files into profiles/ afterwards.
# Kernel timings (--kernel-trace --stats) and PMC counters (--pmc) are collected in SEPARATE rocprofv3 runs.
set -u
TAG=${1:-r03}
shift || true
SECTIONS=${*:-step codec prefill pmc attn stream gemm}      # optional: only these sections (+ "probes", round 5; "pmcsq", round 6)
want() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
W=/tmp/prof_$TAG
rm -rf "$W"; mkdir -p "$W"
HEAD=$(git rev-parse --short HEAD 2>/dev/null || echo "snapshot")
SHA=$(sha1sum fish_speech_amd/libfishmi.so | cut -c1-12)

run_stats() {  # name, header, command...
  local name=$1 hdr=$2; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$W/$name" -o "$name" -- "$@" ) > "$OUT/${name}_run.log" 2>&1
  local db; db=$(find "$W/$name" -name '*_results.db' | head -1)
  { echo "# $hdr"; echo "# libfishmi.so sha1 $SHA, tree $HEAD"; grep -h "decode_frame_avg\|planes=\|prefill of 8\|encode B" "$OUT/${name}_run.log" | sed 's/^/# /' | cut -c1-400;
    python "$PWD/tools/rocpd_summary.py" "$db"; } > "$OUT/${TAG}_kernels_${name}.txt" 2>&1
}

R=$PWD
want step && run_stats step "python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline under rocprofv3 --kernel-trace --stats" \
  python "$R/bench.py" --steps 2 --warmup 1 --no-extras --no-cpu-baseline
want codec && run_stats codec "tools/codec_bench.py (B=8, T=215, fp16-split default + encode) under rocprofv3 --kernel-trace --stats" \
  env PLANES=2 python "$R/tools/codec_bench.py"
want prefill && run_stats prefill "tools/prefill_bench.py 200 2048 (S2-Pro shape, 8 prompts) under rocprofv3 --kernel-trace --stats" \
  python "$R/tools/prefill_bench.py" 200 2048

# HBM traffic of the decode frame: PMC pass on its own (12 frames: 1 prefill frame + 11 graph-replayed decode frames)
if want pmc; then
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$W/pmc" -o p -- \
    python "$R/bench.py" --frames 12 --steps 1 --warmup 0 --no-codec --no-extras --no-cpu-baseline ) > "$OUT/pmc_run.log" 2>&1
CSV=$(find "$W/pmc" -name '*counter_collection.csv' | head -1)
{ echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace over bench.py --frames 12 --steps 1 --warmup 0 --no-codec (libfishmi.so sha1 $SHA, tree $HEAD)";
  python tools/pmc_traffic.py "$CSV" 12 "$OUT/pmc_traffic.json"; } > "$OUT/${TAG}_pmc_fetch_decode.txt" 2>&1
fi

# decode attention at long contexts: one kernel-trace run per context length
if want attn; then
{ echo "# decode attention at B = 8, S2-Pro shape over the context length: attn_decode_fused_kernel (VALU, rows below the threshold position) / attn_decode_mfma_kernel + attn_decode_merge_kernel (rows at or beyond it); tools/attn_decode_probe.py T under rocprofv3 --kernel-trace --stats";
  echo "# libfishmi.so sha1 $SHA, tree $HEAD; bytes per launch = 32768 x S; 'GB/s' = bytes / avg duration"; } > "$OUT/${TAG}_attn_decode.txt"
for T in 300 1024 2048; do
  # (FMI_ATTN_THR=512 so that the MFMA pair is also measured at 1 k keys, below the shipped threshold of 1024)
  ( cd /tmp && FMI_ATTN_THR=${ATTN_THR:-512} rocprofv3 --kernel-trace --stats -d "$W/attn$T" -o a -- python "$R/tools/attn_decode_probe.py" $T ) > "$OUT/attn${T}_run.log" 2>&1
  db=$(find "$W/attn$T" -name '*_results.db' | head -1)
  grep -h "^T=" "$OUT/attn${T}_run.log" >> "$OUT/${TAG}_attn_decode.txt"
  python - "$db" $T >> "$OUT/${TAG}_attn_decode.txt" <<'PY'
import sqlite3, sys
db, T = sys.argv[1], int(sys.argv[2])
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%attn_decode%' group by name order by name"))
for name, n, avg, mn in rows:
    extra = ""
    if "attn_decode_fused" in name or "attn_decode_mfma" in name:
        b = 32768 * (T + 10)
        extra = f"   <- {b / 1e6:.1f} MB per launch = {b / avg / 1e3:.0f} GB/s"
    print(f"  {avg:8.2f} us avg ({mn:.2f} min) x{n:5d}  {name[:90]}{extra}")
PY
done
fi

# SQ counters (round 6): what the waves of the dominant kernels do with their cycles -- the decode GEMVs park on memory
# (WAIT_ANY), the prefill GEMM / codec convs show how busy the matrix pipe is.  Own passes (never with --stats / traces
# beyond --kernel-trace); 8 SQ slots per pass.
if want pmcsq; then
sq_pass() {   # name, counters, header, command...
  local name=$1 ctr=$2 hdr=$3; shift 3
  ( cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$W/sq_$name" -o p -- "$@" ) > "$OUT/sq_${name}_run.log" 2>&1
  local csv; csv=$(find "$W/sq_$name" -name '*counter_collection.csv' | head -1)
  { echo "# $hdr"; echo "# rocprofv3 --pmc $ctr (libfishmi.so sha1 $SHA, tree $HEAD); per kernel and grid: counter sums over its dispatches, and the share of SQ_WAVE_CYCLES";
    echo "# WAIT_ANY = wave parked (s_waitcnt / barrier), WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; MFMA_BUSY / BUSY_CYCLES ~ matrix-pipe utilisation of the busy CUs";
    python "$R/tools/pmc_summary.py" "$csv" | head -120; } > "$OUT/${TAG}_pmc_sq_${name}.txt" 2>&1
}
sq_pass decode "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
  "decode frame kernels: bench.py --frames 12 --steps 1 --warmup 0 --no-codec --no-extras --no-cpu-baseline" \
  python "$R/bench.py" --frames 12 --steps 1 --warmup 0 --no-codec --no-extras --no-cpu-baseline
sq_pass codec "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT" \
  "codec decode (B = 8, T = 215, fp16 split): tools/codec_bench.py" env PLANES=2 python "$R/tools/codec_bench.py"
sq_pass prefill "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT" \
  "prefill of 8 x 200 tokens: tools/prefill_bench.py 200" python "$R/tools/prefill_bench.py" 200
fi

# timing-only runs (no profiler): streaming breakdown + latency
want stream && python tools/stream_breakdown.py > "$OUT/${TAG}_stream_breakdown.txt" 2>&1
want stream && python tools/stream_latency.py > "$OUT/${TAG}_stream_latency.txt" 2>&1
# prefill GEMM variants per shape (tools/gemm_bench.hip; the binary is built on the authoring side: see the file's header)
if want gemm && [ -x tools/bin/gemm_bench ]; then
{ echo "# tools/bin/gemm_bench wxptsua on MI355X (libfishmi.so sha1 $SHA, tree $HEAD): w = wave-specialised 128x128 (round 3), x = 128x256 three-stage tile (round 3), p / u / s / t = linear_tiled_256p_kernel with 256 / 192 / 128 / 64-row tiles (round 4: what launch_linear_tiled chooses among); a = what the library selects (round 6: w2 with the contraction split in three + a reduce pass); every other variant checked bit for bit against the 4-wave LDS-staged kernel"; tools/bin/gemm_bench wxptsua; } > "$OUT/${TAG}_gemm_bench.txt" 2>&1
fi
# round-5 probes (binaries from tools/build_tools.sh; each prints its own reading): sampler stages, the K-slice GEMV
# decomposition, Infinity-Cache residency, prefetch from the previous kernel; the overlapped step against the serial one
if want probes; then
  [ -x tools/bin/sampler_bench ] && { echo "== bucket counts (default)"; tools/bin/sampler_bench; echo "== FMI_SAMPLE_DESCENT=1"; FMI_SAMPLE_DESCENT=1 tools/bin/sampler_bench; } > "$OUT/${TAG}_sampler_stages.txt" 2>&1
  [ -x tools/bin/gemv_ksplit_probe ] && tools/bin/gemv_ksplit_probe > "$OUT/${TAG}_gemv_ksplit_probe.txt" 2>&1
  [ -x tools/bin/mall_resident_probe ] && tools/bin/mall_resident_probe > "$OUT/${TAG}_mall_resident_probe.txt" 2>&1
  [ -x tools/bin/l2_prefetch_probe ] && tools/bin/l2_prefetch_probe > "$OUT/${TAG}_l2_prefetch_probe.txt" 2>&1
  python tools/overlap_step_probe.py 4 plain,prio,floor > "$OUT/${TAG}_overlap_step.txt" 2>&1
fi
ls -la "$OUT"
