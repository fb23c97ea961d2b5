export TMPDIR=/tmp
out=gpurun_out/stall; mkdir -p $out
n=0
for set in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum" "MemUnitStalled VALUBusy GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
n=$((n+1))
timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/p$n -- python3 bench.py --workload quicked --no-cpu-baseline --no-e2e --no-strong --steps 1 --warmup 0 --sync-each-step > $out/log$n.txt 2>&1
cp $out/p$n/*/*counter_collection.csv $out/set$n.csv
rm -rf $out/p$n
done
