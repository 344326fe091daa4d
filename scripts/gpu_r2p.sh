#!/bin/bash
# round 2, call p: batched pipeline throughput at one GPU + the 40B Q4_K single-GPU line
mkdir -p gpurun_out/r2p
cd /root/repo
for b in 1 2 4; do
  timeout 600 python bench.py --force-pipeline --streams 2 --pipe-batch $b --steps 64 --warmup 8 > gpurun_out/r2p/pipe_7b_b$b.json 2> gpurun_out/r2p/pipe_7b_b$b.err
done
timeout 600 python bench.py --force-pipeline --streams 4 --pipe-batch 4 --steps 64 --warmup 8 > gpurun_out/r2p/pipe_7b_g4b4.json 2> gpurun_out/r2p/pipe_7b_g4b4.err
( time timeout 900 python bench.py --model 40b --quant q4_k --no-cpu --prefill-long 0 --steps 32 --warmup 4 --repeats 1 ) > gpurun_out/r2p/bench_40b_q4k.json 2> gpurun_out/r2p/bench_40b_q4k.err
tail -c 600 gpurun_out/r2p/*.json
tail -5 gpurun_out/r2p/*.err
