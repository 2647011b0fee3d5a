#!/bin/bash
# Round-6 counter collection at HEAD (run through gpurun from the repo root): one bench line on the same box, then separate rocprofv3 --pmc passes
# (never combined with anything but --kernel-trace), C4 and north-star for the traffic counters, C4 + the pds-class LP for the pipe / LDS counters.
# Results land in gpurun_out/${TAG}_*; the summaries are copied into profiles/ by hand.  TAG defaults to r06.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG=${TAG:-r06}
O=gpurun_out/${TAG}
nproc > ${O}_host.txt; grep -m1 "model name" /proc/cpuinfo >> ${O}_host.txt
if [ -z "$NO_BENCH" ]; then timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err; fi
B="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3 --unpaired"
for wl in c4 headline; do
  for c in FETCH_SIZE WRITE_SIZE; do
    TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d ${O}_pmc_${wl}_$c -- python bench.py --workload $wl $B > ${O}_pmc_${wl}_$c.log 2>&1
  done
  python tools/pmc_to_json.py $wl ${O}_pmc_${wl}_FETCH_SIZE ${O}_pmc_${wl}_WRITE_SIZE ${O}_pmc_k_update.json \
    "round 6 (${TAG}): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes), TLPK_STREAMS=1 TLPK_SERIAL=1 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3 --unpaired (tools/collect_pmc_r06.sh); FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md section HBM; calibration profiles/r01_pmc_k_update_hbm_traffic.md); NOT collected in the bench run itself" >> ${O}_pmc_to_json.log 2>&1
done
for wl in c4 pds; do
  TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d ${O}_pmc_${wl}_mfma -- python bench.py --workload $wl $B > ${O}_pmc_${wl}_mfma.log 2>&1
  TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d ${O}_pmc_${wl}_lds -- python bench.py --workload $wl $B > ${O}_pmc_${wl}_lds.log 2>&1
done
python tools/pmc_summarise.py ${O}_pmc_c4_FETCH_SIZE ${O}_pmc_c4_WRITE_SIZE ${O}_pmc_c4_mfma ${O}_pmc_c4_lds > ${O}_pmc_summary.md 2>&1
python tools/pmc_summarise.py ${O}_pmc_headline_FETCH_SIZE ${O}_pmc_headline_WRITE_SIZE > ${O}_pmc_summary_headline.md 2>&1
python tools/pmc_summarise.py ${O}_pmc_pds_mfma ${O}_pmc_pds_lds > ${O}_pmc_summary_pds.md 2>&1
rm -rf ${O}_pmc_*_FETCH_SIZE ${O}_pmc_*_WRITE_SIZE ${O}_pmc_*_mfma ${O}_pmc_*_lds
cat ${O}_pmc_to_json.log; head -c 600 ${O}_bench.json; echo; head -30 ${O}_pmc_summary.md
