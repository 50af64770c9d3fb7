#!/bin/bash
# A round's measurements on one MI355X box:   gpurun --timeout 3300 -- bash tools/measure_round.sh r06
# bench lines (fp32 with the fp32-pipe A/B, fp16 activations), rocprofv3 kernel statistics of the model step and of an
# interior rank of 8 with their per-category splits, PMC passes (splat traffic, 1x1 and 3x3 kernels), per-rank cost of the
# sharded step in every transport, the 8-rank partition as communicating processes on this one GPU, fuzz sweeps.
# Everything lands in gpurun_out/<tag>/ ; copy what is to be kept into profiles/.
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
( time timeout 1500 python bench.py > $o/${tag}_bench.json 2> $o/bench.err ); echo "bench rc=$?"
head -c 300 $o/${tag}_bench.json; echo
timeout 900 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null
timeout 900 python bench.py --fp16-activations --no-cpu-baseline --no-stages > $o/train_fp16.json 2>/dev/null
cat $o/infer32_fp16.json $o/train_fp16.json | grep '^{' > $o/${tag}_bench_fp16_activations.jsonl
python - <<PY
import json
for l in open("$o/${tag}_bench_fp16_activations.jsonl"):
    d = json.loads(l); print("fp16 activations:", d["config"]["workload"][:40], d["value"], d["ms_per_step"])
PY
timeout 900 python scripts/bench_ops.py > $o/${tag}_bench_ops.jsonl 2>/dev/null
# per-rank cost of the sharded 720p step: through the IPC mailboxes to self (default), exchanges stubbed, over RCCL to self
( for mode in "--ipc-self" "" "--rccl-self"; do timeout 400 python tools/rank_cost.py $mode 8 2>&1 | grep "^world"; done
  timeout 400 python tools/rank_cost.py --ipc-self 4 2>&1 | grep "^world"
  timeout 600 python tools/rank_cost.py --ipc-self --4k 8 2>&1 | grep "^world" | sed "s/$/ [3840x2160]/"
  timeout 600 python tools/rank_cost.py 1 2 2>&1 | grep "^world"
  SBMC_WBANK=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [SBMC_WBANK=0]/"
  SBMC_HIP_PW_GW_WIDE=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [SBMC_HIP_PW_GW_WIDE=0]/"
  SBMC_POOL_SKIP=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [SBMC_POOL_SKIP=0]/"
  SBMC_CONV3X3_STREAMK=0 timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s/$/ [SBMC_CONV3X3_STREAMK=0]/"
  # round 6's 3x3 forms off: the one-wave-per-SIMD forward / data gradient, the four-wave weight gradient
  SBMC_CONV3X3_WS=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [SBMC_CONV3X3_WS=0]/"
  SBMC_CONV3X3_WGRAD_W8=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [SBMC_CONV3X3_WGRAD_W8=0]/"
  SBMC_CONV3X3_WS=0 SBMC_CONV3X3_WGRAD_W8=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [SBMC_CONV3X3_WS=0 SBMC_CONV3X3_WGRAD_W8=0]/"
  timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" ) | tee $o/${tag}_rank_cost.txt
# the real 8-rank partition as 8 communicating processes on this one GPU (gloo collectives, IPC mailboxes)
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 2 --no-cpu-baseline --no-stages > $o/${tag}_bench_8ranks_one_gpu.json 2> $o/bench8.err; echo "8 ranks on one GPU rc=$?"
head -c 400 $o/${tag}_bench_8ranks_one_gpu.json; echo
# rocprofv3: splat + model kernel statistics, PMC passes of the splat kernels (-> <tag>_pmc.json: the bench line's traffic)
bash tools/prof.sh $tag > $o/prof.log 2>&1; echo "prof rc=$?"
cp gpurun_out/profiles_$tag/* $o/ 2>/dev/null
python tools/prof_rank_cat.py $o/${tag}_model_kernel_stats.csv 11 > $o/${tag}_model_categories.txt; head -14 $o/${tag}_model_categories.txt
rm -rf gpurun_out/prof_$tag gpurun_out/profiles_$tag        # (raw traces: gpurun brings back 64 MiB at most)
# an interior rank of 8, every exchange running: kernel statistics + categories
bash tools/prof_rank.sh 8 --ipc-self > $o/prof_rank.log 2>&1; cp gpurun_out/q/rank8_stats.csv $o/${tag}_rank8_kernel_stats.csv; rm -rf gpurun_out/q
python tools/prof_rank_cat.py $o/${tag}_rank8_kernel_stats.csv 5 > $o/${tag}_rank8_categories.txt; head -18 $o/${tag}_rank8_categories.txt
bash tools/prof_pointwise.sh > $o/prof_pw.log 2>&1; cp gpurun_out/profiles_pw/r02_pointwise_pmc.txt $o/${tag}_pointwise_pmc.txt; grep -v raw $o/${tag}_pointwise_pmc.txt | tail -5 | cut -c1-260; rm -rf gpurun_out/prof_pw gpurun_out/profiles_pw
timeout 300 python tools/bench_pw_scaled.py > $o/${tag}_pointwise_launches.txt 2>&1; tail -3 $o/${tag}_pointwise_launches.txt
# round 6: the fused 1x1 chains beside the layers they replace (+ cycles per phase), the wide forward, the pair backward
( SBMC_PC_TIMING=1 timeout 300 python tools/bench_pw_chain.py 2>&1 | grep -v amdgpu.ids; timeout 200 python tools/dev/bench_wide_fwd.py 2>&1 | grep SBMC; timeout 300 python tools/dev/bench_chain_bwd.py 2>&1 | grep -v amdgpu.ids ) > $o/${tag}_pointwise_chain.txt; cat $o/${tag}_pointwise_chain.txt
SBMC_PW_CHAIN=0 SBMC_PW_WIDE_FWD=0 timeout 500 python bench.py --no-cpu-baseline --no-stages > $o/${tag}_bench_without_chains.json 2>/dev/null; head -c 200 $o/${tag}_bench_without_chains.json; echo
bash tools/calibrate_fetch.sh > /dev/null 2>&1; cp gpurun_out/fetch_calibration.txt $o/${tag}_fetch_calibration.txt
bash tools/prof_conv_stack.sh > $o/prof_conv.log 2>&1; cp gpurun_out/conv_stack/r02_conv_stack_mfma.txt $o/${tag}_conv_stack_mfma.txt; tail -3 $o/${tag}_conv_stack_mfma.txt; rm -rf gpurun_out/conv_stack
( timeout 300 python tools/fuzz_gpu.py --seconds 100 2>&1 | tail -2; timeout 300 python tools/fuzz_slab.py --seconds 100 2>&1 | tail -1; timeout 200 python tools/fuzz_pointwise.py --seconds 60 2>&1 | tail -1; timeout 200 python tools/fuzz_pointwise_chain.py --seconds 40 2>&1 | tail -1 ) > $o/${tag}_fuzz.txt; cat $o/${tag}_fuzz.txt | cut -c1-300
timeout 400 python tools/conv3x3_experiment.py --shapes all > $o/${tag}_conv3x3_experiment.txt 2>&1; tail -2 $o/${tag}_conv3x3_experiment.txt
# the same launches on operands that toggle no bits (what the power limit has to do with the time), both forms of each kernel
( for ws in 1 0; do echo "SBMC_CONV3X3_WS=$ws SBMC_CONV3X3_WGRAD_W8=$ws"; SBMC_CONV3X3_WS=$ws SBMC_CONV3X3_WGRAD_W8=$ws timeout 300 python tools/conv3x3_experiment.py --power --reps 20 2>&1 | grep -v "values\|adjoint\|amdgpu.ids"; done ) > $o/${tag}_conv3x3_power.txt; cat $o/${tag}_conv3x3_power.txt
bash tools/prof_conv3x3.sh all > $o/prof_conv3.log 2>&1; cp gpurun_out/conv3x3_pmc.txt $o/${tag}_conv3x3_pmc.txt; grep "sbmc::conv3" $o/${tag}_conv3x3_pmc.txt | cut -c1-200
timeout 900 python tools/fuzz_conv3x3.py --cases 400 2>&1 | tail -1 > $o/${tag}_conv3x3_fuzz.txt; cat $o/${tag}_conv3x3_fuzz.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -f $o/*.err $o/prof*.log; du -sh gpurun_out
