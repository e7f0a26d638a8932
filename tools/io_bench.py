"""Host throughput of libtfr_io.so's ELWC parser (no GPU): lists/s and MB/s per thread count.
   Layouts: 'wide'  = one float feature of width 136 per example (packed),
            'scalar' = 136 scalar float features "1".."136" per example (the reference's examples/data layout).
   Every line twice: fp32 example features, and bf16 features with the label kept fp32 (bf16 ingest).
   usage: python tools/io_bench.py [--lists 256] [--list-size 100]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import data_ref as D                     # encoder only (test infrastructure; this is a tool)
from ranking_amd import data as rd


def make(layout, n_lists, L, F, seed=0):
    rng = np.random.RandomState(seed)
    recs = []
    for _ in range(n_lists):
        exs = []
        for _ in range(L):
            x = rng.uniform(-1, 1, F).astype(np.float32)
            feats = {'label': ('float', [float(rng.randint(0, 5))])}
            if layout == 'wide':
                feats['x'] = ('float', x.tolist())
            else:
                for k in range(F):
                    feats[str(k + 1)] = ('float', [float(x[k])])
            exs.append(feats)
        recs.append(D.encode_elwc(None, exs))
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lists', type=int, default=128)
    ap.add_argument('--list-size', type=int, default=100)
    ap.add_argument('--features', type=int, default=136)
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    for layout in ('wide', 'scalar'):
        recs = make(layout, a.lists, a.list_size, a.features)
        nbytes = sum(len(r) for r in recs)
        spec = {'label': rd.FixedLenFeature([1], torch.float32, default_value=-1.0)}
        if layout == 'wide':
            spec['x'] = rd.FixedLenFeature([a.features], torch.float32, default_value=0.0)
        else:
            for k in range(a.features):
                spec[str(k + 1)] = rd.FixedLenFeature([1], torch.float32, default_value=0.0)
        for th in (1, 2, 4, 8):
            for tag, kw in (('fp32', {}), ('bf16', {'example_dtype': torch.bfloat16, 'float32_features': ('label',)})):
                best = 1e9
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    rd.parse_from_example_list(recs, list_size=a.list_size, example_feature_spec=spec, num_threads=th,
                                               **kw)
                    best = min(best, time.perf_counter() - t0)
                print('%-6s %s threads=%d  %8.0f lists/s  %8.1f MB/s  (%d lists, %.1f MB)' % (
                    layout, tag, th, a.lists / best, nbytes / best / 1e6, a.lists, nbytes / 1e6))


if __name__ == '__main__':
    main()
