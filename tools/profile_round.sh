#!/bin/bash
# Everything the round's committed profiles/ come from, in one GPU call (run from the repo root on the GPU box):
#   bench lines for every BASELINE workload, rocprofv3 kernel stats of the default command and of the every-kernel-alone command,
#   FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only beside --pmc) of the every-kernel-alone command,
#   two SQ counter passes over the encoder-shaped attention launch (tools/attn_pmc.py).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --kernels --cached-refs > $O/bench.json 2> $O/bench.err || exit 1
for w in cfg3 cfg4 cfg5; do python3 $R/bench.py --workload $w --no-cfg4 --no-eager --no-cpu-baseline --steps 8 > $O/bench_$w.json 2> $O/bench_$w.err || exit 1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --steps 10 > $O/stats_default.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_alone -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --inflight 1 --lanes 1 --chunk 48 --steps 10 > $O/stats_alone.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --inflight 1 --lanes 1 --chunk 48 --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-eager --no-cfg4 --inflight 1 --lanes 1 --chunk 48 --steps 3 --warmup 1 > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_attn1 -- python3 $R/tools/attn_pmc.py > $O/pmc_attn1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_attn2 -- python3 $R/tools/attn_pmc.py > $O/pmc_attn2.log 2>&1 || exit 1
cd $R && python3 tools/summarise_attn_pmc.py $O/attn_pmc.json $O/pmc_attn1 $O/pmc_attn2 > /dev/null || exit 1
cd $R && python3 tools/summarise_prof.py $O/stats_alone $O/pmc_fetch $O/pmc_write $O/alone && python3 tools/summarise_prof.py $O/stats_default $O/pmc_fetch $O/pmc_write $O/default
echo done
