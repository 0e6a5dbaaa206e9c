set -u
export TMPDIR=/tmp
out=gpurun_out/pmc1
mkdir -p $out
python tools/conv_variants.py resnet > $out/variants.txt 2>&1
ARGS="8 50 84 1024 256 1 1 g128x128k64"
rm -rf $out/a $out/b $out/c
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/a -- python tools/pmc_one.py $ARGS > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python tools/pmc_one.py $ARGS > $out/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/c -- python tools/pmc_one.py $ARGS > $out/c.log 2>&1
for d in a b c; do python tools/pmc_summary.py $out/$d conv >> $out/pmc.txt; done
rm -rf $out/a $out/b $out/c
cat $out/variants.txt; cat $out/pmc.txt
