#!/usr/bin/env python3
"""Tiny dynamics-only workload for rocprofv3 counter collection (E episodes, T steps, L lanes/wave)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
E, T, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eng = serl_amd.RolloutEngine(0)
cmds = np.zeros((E, T, 10)); cmds[:, :, 0] = 0.01 * np.sin(np.arange(T) * 0.01)[None]
eng.dynamics_open_loop(cmds, lanes_per_wave=L)
print('ms', eng.last_kernel_ms, 'us/step', eng.last_kernel_ms * 1e3 / T)
