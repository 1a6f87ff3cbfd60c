# Round-end measurement bundle, part 2 (GPU box): rocprofv3 --pmc passes, ONE counter group per run (FETCH_SIZE and
# WRITE_SIZE together exceed what one pass can collect: rocprofv3 aborts and then hangs in finalisation), each under timeout.
# usage: bash tools/final_pmc.sh [rN] [commit]
export TMPDIR=/tmp
R=${1:-r5}; C=${2:-unknown}; O=gpurun_out/final; mkdir -p $O
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-mask --no-feeds --no-configs"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m /tmp/pmc_fw
timeout 150 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f --output-format csv -- $B > $O/pmc_f.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w --output-format csv -- $B > $O/pmc_w.log 2>&1
timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_m --output-format csv -- $B > $O/pmc_m.log 2>&1
mkdir -p /tmp/pmc_fw; cp -r /tmp/pmc_f /tmp/pmc_fw/f 2>/dev/null; cp -r /tmp/pmc_w /tmp/pmc_fw/w 2>/dev/null
python tools/pmc_summary.py /tmp/pmc_fw $O/${R}_pmc_fetch_write_summary.json $C > $O/pmc_sum.log 2>&1
python tools/pmc_summary.py /tmp/pmc_m $O/${R}_pmc_mfma_summary.json $C >> $O/pmc_sum.log 2>&1
