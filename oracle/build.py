"""Build the C restatement (oracle/_build/libcitation_oracle.so).  Test infrastructure."""
import os, subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libcitation_oracle.so')


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ('citation_ref.c', 'rollout_ref.c', 'citation_rt.h', 'citation_step.h',
                                            '../include/serl_amd.h')]
    srcs += [os.path.join(HERE, 'gen', f) for f in sorted(os.listdir(os.path.join(HERE, 'gen')))]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    subprocess.run(['make', '-C', HERE, '-s'], check=True)
    return LIB


def build_xcheck(force=False):
    """GPU kernels around the lifted model (oracle/xcheck/): the checker's second implementation on the device."""
    from oracle import xcheck
    return xcheck.build(force)


if __name__ == '__main__':
    print(build(force=True))
    print(build_xcheck(force=True))
