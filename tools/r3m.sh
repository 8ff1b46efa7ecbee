#!/bin/bash
python -m pytest tests -m gpu -q -x > gpurun_out/r3m_tests.log 2>&1; tail -3 gpurun_out/r3m_tests.log
tools/ab.sh "--kout 1 --what sil --iters 30" - nobl > gpurun_out/r3m_ab.txt 2>&1; cat gpurun_out/r3m_ab.txt
python bench.py --steps 20 --warmup 5 --cpu-seconds 5 > gpurun_out/r3m_bench2.json 2> gpurun_out/r3m_bench2.err
python bench.py --config 4 --steps 20 --warmup 5 --cpu-seconds 3 > gpurun_out/r3m_bench4.json 2> gpurun_out/r3m_bench4.err
python bench.py --config 3 --steps 20 --warmup 5 --cpu-seconds 3 > gpurun_out/r3m_bench3.json 2> gpurun_out/r3m_bench3.err
