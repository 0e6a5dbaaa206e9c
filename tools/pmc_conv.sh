# usage: tools/pmc_conv.sh "<n H W cin cout k s hint>" <tag>   (on the GPU box)
set -u
export TMPDIR=/tmp
ARGS="$1"; tag=${2:-x}
out=gpurun_out/pmc_$tag
mkdir -p $out; rm -f $out/pmc.txt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/a -- python tools/pmc_one.py $ARGS > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python tools/pmc_one.py $ARGS > $out/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $out/c -- python tools/pmc_one.py $ARGS > $out/c.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_LEVEL_WAVES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/d -- python tools/pmc_one.py $ARGS > $out/d.log 2>&1
for d in a b c d; do python tools/pmc_summary.py $out/$d conv >> $out/pmc.txt 2>/dev/null; done
rm -rf $out/a $out/b $out/c $out/d
echo "== $ARGS"; cat $out/pmc.txt
