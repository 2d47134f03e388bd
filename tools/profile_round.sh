#!/bin/bash
# Everything the round's committed profiles/ come from, in one GPU call (run from the repo root on the GPU box):
#   tools/profile_round.sh r05     ->  gpurun_out/prof_r05/...   (copy the summaries into profiles/r05_* afterwards: tools/collect_profiles.py r05)
#   bench lines for every BASELINE workload (+ the bf16 operand mode of cfg-2), rocprofv3 kernel stats of the default command and of the
#   every-kernel-alone command (cfg-2 and cfg-4), the kernel-concurrency timeline of the default command, FETCH_SIZE / WRITE_SIZE PMC
#   passes (separate runs, --kernel-trace only beside --pmc) of the every-kernel-alone commands, two SQ counter passes over the
#   encoder-shaped attention launch (tools/attn_pmc.py), the GEMM kernels against hipBLASLt, what the chip sustains on bare MFMA loops
#   (tools/mfma_peak.py) and the phase clocks of the 256 x 256 x 64 GEMM (tools/gemm256_phases.py).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --kernels --cached-refs > $O/bench.json 2> $O/bench.err || exit 1
python3 $R/bench.py --dtype bf16 --no-eager --no-cfg4 --no-more-configs > $O/bench_bf16.json 2> $O/bench_bf16.err || exit 1
for w in cfg3 cfg4 cfg5; do python3 $R/bench.py --workload $w --no-cfg4 --no-more-configs --no-eager --no-cpu-baseline --steps 8 > $O/bench_$w.json 2> $O/bench_$w.err || exit 1; done
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --steps 10 > $O/stats_default.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_alone -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --inflight 1 --lanes 1 --chunk 48 --steps 10 > $O/stats_alone.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_alone_cfg4 -- python3 $R/bench.py --workload cfg4 --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --inflight 1 --lanes 1 --chunk 96 --steps 4 > $O/stats_alone_cfg4.log 2>&1 || exit 1
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --inflight 1 --lanes 1 --chunk 48 --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --inflight 1 --lanes 1 --chunk 48 --steps 3 --warmup 1 > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_cfg4 -- python3 $R/bench.py --workload cfg4 --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --inflight 1 --lanes 1 --chunk 96 --steps 2 --warmup 1 > $O/pmc_fetch_cfg4.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_cfg4 -- python3 $R/bench.py --workload cfg4 --no-cpu-baseline --no-eager --no-cfg4 --no-more-configs --no-repeats --inflight 1 --lanes 1 --chunk 96 --steps 2 --warmup 1 > $O/pmc_write_cfg4.log 2>&1 || exit 1
echo "traffic passes done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_attn1 -- python3 $R/tools/attn_pmc.py > $O/pmc_attn1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_attn2 -- python3 $R/tools/attn_pmc.py > $O/pmc_attn2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_k1 -- python3 $R/tools/panel_pmc.py > $O/pmc_k1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_k2 -- python3 $R/tools/panel_pmc.py > $O/pmc_k2.log 2>&1 || exit 1
# the four-wave token-panel kernel (csrc/panel4.hip, opt-in): alone (both kernels, one process, alternating), its SQ counters and its phase clocks
cd $R && python3 tools/panel4_ab.py > $O/panel4_ab.log 2>&1 || exit 1
cd /tmp && CS_PANEL_M=65760 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_panel4 -- python3 $R/tools/panel4_pmc.py > $O/stats_panel4.log 2>&1 || exit 1
CS_PANEL_M=65760 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_p4a -- python3 $R/tools/panel4_pmc.py > $O/pmc_p4a.log 2>&1 || exit 1
CS_PANEL_M=65760 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $O/pmc_p4b -- python3 $R/tools/panel4_pmc.py > $O/pmc_p4b.log 2>&1 || exit 1
cd $R && python3 tools/pmc_quick.py $O/pmc_p4a $O/pmc_p4b > $O/panel4_pmc.txt 2>&1 || exit 1
cd $R && if [ -d tools/_var/p4stamp ]; then CS_VARIANT=p4stamp python3 tools/panel4_phases.py > $O/panel4_phases.txt 2>&1 || exit 1; fi
echo "attention, panel and GEMM counters done"
cd $R && python3 tools/summarise_attn_pmc.py $O/attn_pmc.json $O/pmc_attn1 $O/pmc_attn2 > /dev/null || exit 1
cd $R && python3 tools/summarise_kernel_pmc.py $O/kernel_pmc.json $O/pmc_k1 $O/pmc_k2 > /dev/null || exit 1
cd $R && python3 tools/summarise_prof.py $O/stats_alone $O/pmc_fetch $O/pmc_write $O/alone && python3 tools/summarise_prof.py $O/stats_default $O/pmc_fetch $O/pmc_write $O/default \
  && python3 tools/summarise_prof.py $O/stats_alone_cfg4 $O/pmc_fetch_cfg4 $O/pmc_write_cfg4 $O/alone_cfg4 || exit 1
cd $R && python3 tools/timeline.py $O/stats_default $O/timeline.json > $O/timeline.txt 2>&1 || exit 1
cd $R && CS_GVB_IMGS=96,22,8 python3 tools/gemm_vs_blas.py > $O/gemm_vs_blas.log 2>&1 || exit 1
cd $R && python3 tools/mfma_peak.py > $O/mfma_peak.log 2>&1 || exit 1
cd $R && python3 tools/gemm256_phases.py > $O/gemm256_phases.log 2>&1 || exit 1
echo done
