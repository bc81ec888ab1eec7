set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_final
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 tools/prof_eval.py 512 1 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 tools/prof_eval.py 512 1 > $O/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/va -o a -- python3 tools/prof_eval.py 512 1 > $O/va.log 2>&1
echo valu A done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/vb -o b -- python3 tools/prof_eval.py 512 1 > $O/vb.log 2>&1
echo valu B done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mf -o m -- python3 tools/prof_eval.py 512 1 > $O/mf.log 2>&1
echo mfma done
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
A=$(find $O/va -name "*counter_collection.csv" | head -1); B=$(find $O/vb -name "*counter_collection.csv" | head -1); M=$(find $O/mf -name "*counter_collection.csv" | head -1)
python3 tools/make_traffic_json.py $F $W 512 $O/traffic.json
python3 tools/pmc_valu.py $A $B gram_kernel trace_kernel trace_group > $O/valu.csv
python3 tools/pmc_clock.py $M 300 > $O/mfma_busy_by_launch.csv
# keep only the summaries (the raw CSVs are large)
rm -rf $O/fetch $O/write $O/va $O/vb $O/mf
ls -la $O
