#!/bin/bash
# PASSES="q3 q4 q6" selects passes.
# round 6: issue / stall / queue / unit counters of the strip up-sampling kernel on ONE layer (LAYER=0: SR block 1 conv0, 8 views bf16), and
# of round 5's overlapping-tile kernel beside it (NFE_UP_STRIP=0):   tools/r06_up_pmc.sh [math] [views]
export TMPDIR=/tmp
MATH=${1:-bf16}; NV=${2:-8}
for form in strips tiles; do
  OUT=gpurun_out/r06_up_pmc_$form
  [ -z "$KEEP" ] && rm -rf $OUT; mkdir -p $OUT
  pass() {
    name=$1; shift
    case " ${PASSES:-q1 q2 q3 q4 q6} " in *" $name "*) ;; *) return;; esac
    ( [ $form = tiles ] && export NFE_UP_STRIP=0; ITERS=6 LAYER=${LAYER:-0} timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/time_up.py $MATH $NV > $OUT/$name.log 2>&1 )
    echo "pass $form $name rc=$?"
  }
  pass q1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_WAVES
  pass q2 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
  pass q3 TA_TA_BUSY TA_FLAT_WRITE_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS TCP_TOTAL_CACHE_ACCESSES      # (round 6: a pass with TA_*_STALLED_BY_TC_CYCLES / TA_FLAT_READ_LDS_WAVEFRONTS aborted inside rocprofv3 and hung: every pass runs under `timeout` now)
  pass q4 FETCH_SIZE WRITE_SIZE TCC_HIT TCC_MISS TCP_TCC_READ_REQ GRBM_GUI_ACTIVE
  pass q6 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
  if [ $form = strips ]; then K="upconv_strip_kernel"; else K="conv3_kernel<1, 1, true"; fi
  PMC_KERNEL="$K" python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
  grep -A60 "== void nfe::$K" $OUT/summary.txt | head -64
done
