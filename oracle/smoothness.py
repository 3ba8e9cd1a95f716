"""Restatement of base/core/utils.py:39-58 (calc_nMAE) and :82-120 (calc_smoothness).  TEST
INFRASTRUCTURE.  Uses numpy.fft (the reference uses scipy.fftpack.fft; same DFT)."""
import numpy as np


def calc_smoothness(y, dt=0.01):
    y = np.asarray(y, dtype=np.float64)
    N, A = y.shape
    T = N * dt
    freq = np.linspace(dt, 1 / (2 * dt), N // 2 - 1)
    Syy = np.zeros((N // 2 - 1, A))
    for i in range(A):
        Y = np.fft.fft(y[:, i], N)
        Syy[:, i] = np.abs(Y[1:N // 2] * np.conjugate(Y[1:N // 2])) * dt
    rough = np.einsum('ij,i->j', Syy, freq) * 2 / N
    return -(np.sqrt(np.sum(rough, axis=-1)) * 100 * (80 / T))


def calc_nMAE(error):
    error = np.asarray(error)
    mae = np.mean(np.absolute(error), axis=0)
    rng = np.array([np.deg2rad(20), np.deg2rad(20), max(np.abs(np.average(error[:, -1])), 3.14159 / 180)])
    return np.mean(mae / rng) * 100
