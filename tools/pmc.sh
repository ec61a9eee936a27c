#!/bin/bash
# (round 6: every pass runs under `timeout` - a pass whose counter set rocprofv3 rejects can abort inside the tool and hang until the box limit)
# PMC counter passes for the render kernel (run on the GPU box via gpurun; counters only, no traces
# beyond --kernel-trace).  Usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=gpurun_out/${1:-pmc}; shift || true
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-strong-scaling}
PROG=${PMC_PROG:-bench.py}          # PMC_PROG=tools/time_full.py tools/pmc.sh out 4 128 48 48 bf16x3
export TMPDIR=/tmp
mkdir -p $OUT
pass() {
  name=$1; shift
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $PROG $ARGS > $OUT/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU
pass mem1 FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES
pass mem2 WRITE_SIZE TCC_MISS TCC_REQ TCP_TCC_READ_REQ TA_FLAT_READ_WAVEFRONTS
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
