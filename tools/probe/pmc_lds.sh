#!/bin/bash
# LDS bank conflicts + MFMA busy of one conv variant: tools/probe/pmc_lds.sh "<n H W cin cout k s>" hint_name [hint_name ...]
export TMPDIR=/tmp
ARGS="$1"; shift
for h in "$@"; do
  out=/tmp/pmc_$h; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $out/c -- python tools/pmc_one.py $ARGS $h > $out/c.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/b -- python tools/pmc_one.py $ARGS $h > $out/b.log 2>&1
  echo "== $ARGS $h"
  python tools/pmc_summary.py $out/c conv
  python tools/pmc_summary.py $out/b conv
done
