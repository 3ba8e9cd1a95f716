timeout 800 python -m pytest tests/test_gpu_ga.py -q -x -k "distillation or default_config" 2>&1 | tail -30
timeout 600 python tools/gen_timing.py 2 --epoch 2>&1 | tail -2
