#!/usr/bin/env python
"""The canonical example of the reference (tensorflow_ranking/examples/tf_ranking_libsvm.py: LibSVM
data, 136 numeric features, a 2 x 256 feed-forward scorer, pointwise sigmoid cross-entropy or any other
ranking loss, NDCG@{1,3,5,10}) on the MI355X path:

    python examples/train_libsvm.py --train_path train.txt --vali_path vali.txt --loss softmax_loss

Same flag names as the reference where they exist (:78-93)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd as tfr  # noqa: E402


def batches(features, labels, batch_size, shuffle, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = features.shape[0]
    while True:
        order = torch.randperm(n, generator=g) if shuffle else torch.arange(n)
        for lo in range(0, n - (batch_size - 1 if shuffle else 0), batch_size):
            idx = order[lo:lo + batch_size]
            yield {'x': features[idx], 'mask': labels[idx] >= 0}, labels[idx]
        if not shuffle:
            return


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train_path', required=True)
    ap.add_argument('--vali_path', required=True)
    ap.add_argument('--output_dir', default='/tmp/tfr_libsvm')
    ap.add_argument('--train_batch_size', type=int, default=32)          # tf_ranking_libsvm.py:82
    ap.add_argument('--num_train_steps', type=int, default=1000)
    ap.add_argument('--learning_rate', type=float, default=0.01)         # :85
    ap.add_argument('--dropout_rate', type=float, default=0.5)           # :86
    ap.add_argument('--hidden_layer_dims', default='256,128,64')         # :87
    ap.add_argument('--num_features', type=int, default=136)             # :90
    ap.add_argument('--list_size', type=int, default=100)                # :91
    ap.add_argument('--loss', default='sigmoid_cross_entropy_loss')      # :93
    ap.add_argument('--compute_dtype', default='bfloat16', choices=['bfloat16', 'float32'],
                    help='float32 = the reference\'s own tower precision (fp32 MFMA Dense); bfloat16 = the fused tower')
    args = ap.parse_args()

    feats, labels = tfr.data.load_libsvm_data(args.train_path, args.list_size, args.num_features)
    vfeats, vlabels = tfr.data.load_libsvm_data(args.vali_path, args.list_size, args.num_features)
    hidden = [int(d) for d in args.hidden_layer_dims.split(',')]

    class Builder:
        def build(self):
            scorer = tfr.keras.model.DNNScorer(input_dim=args.num_features, hidden_layer_dims=hidden, output_units=1,
                                               activation=torch.relu, use_batch_norm=True, dropout=args.dropout_rate,
                                               compute_dtype=getattr(torch, args.compute_dtype))

            class M(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.scorer = scorer

                def forward(self, f):
                    return self.scorer({}, {'x': f['x']}, f['mask'])
            return M()

    P = tfr.keras.pipeline
    steps_per_epoch = max(1, min(100, args.num_train_steps))
    hp = P.PipelineHparams(model_dir=args.output_dir, num_epochs=max(1, args.num_train_steps // steps_per_epoch),
                           steps_per_epoch=steps_per_epoch, validation_steps=10 ** 9, learning_rate=args.learning_rate,
                           loss=args.loss, optimizer='adagrad', export_best_model=True,
                           best_exporter_metric='metric/ndcg_5', best_exporter_metric_higher_better=True)

    class DS(P.AbstractDatasetBuilder):
        def build_train_dataset(self):
            return batches(feats, labels, args.train_batch_size, True)

        def build_valid_dataset(self):
            return batches(vfeats, vlabels, max(1, min(256, vfeats.shape[0])), False)

    hist = P.SimplePipeline(Builder(), DS(), hp).train_and_validate(verbose=1)
    print({k: round(v[-1], 5) for k, v in hist.items()})


if __name__ == '__main__':
    main()
