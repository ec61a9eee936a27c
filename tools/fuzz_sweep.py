"""One-off wide fuzz: the cases of tests/test_random_sweep_gpu.py drawn from many more seeds (not part of the suite).
    python tools/fuzz_sweep.py [first_seed] [n_forward] [n_backward]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import test_random_sweep_gpu as T  # noqa: E402


def main():
    first, nf, nb = (int(a) for a in (sys.argv[1:4] + ["100", "300", "60"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    bad = []
    for s in range(first, first + nf):
        try:
            T.test_forward_sweep.__wrapped__(s, dev) if hasattr(T.test_forward_sweep, "__wrapped__") else T.test_forward_sweep(s, dev)
        except AssertionError as e:
            bad.append(("fwd", s, str(e)[:200]))
    for s in range(first, first + nb):
        try:
            T.test_backward_sweep(s, dev)
        except AssertionError as e:
            bad.append(("bwd", s, str(e)[:200]))
    print(f"forward {nf} cases, backward {nb} cases from seed {first}: {len(bad)} failures")
    for b in bad[:10]:
        print("  ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
