#!/usr/bin/env python
"""Generates tests/golden/elwc_golden.json from the reference's own data files (run in the build
container, where /root/reference exists): the first records of
examples/data/train_numerical_elwc.tfrecord (base64) with what the pure-Python restatement
(oracle/data_ref.py) decodes from them, plus the head of examples/data/train.txt with the
restated LibSVM loader's output.  The GPU box has no /root/reference: tests read this file."""
import base64
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import data_ref as D  # noqa: E402

REF = '/root/reference/tensorflow_ranking/examples/data'


def main():
    buf = open(os.path.join(REF, 'train_numerical_elwc.tfrecord'), 'rb').read()
    records = D.read_tfrecord(buf)
    take = records[:3]
    decoded = []
    for r in take:
        ctx, exs = D.decode_elwc(r)
        decoded.append({'context': {k: [v[0], [x if not isinstance(x, bytes) else x.decode('latin1') for x in v[1]]]
                                    for k, v in ctx.items()},
                        'examples': [{k: [v[0], [x if not isinstance(x, bytes) else x.decode('latin1') for x in v[1]]]
                                      for k, v in e.items()} for e in exs]})
    text = open(os.path.join(REF, 'train.txt')).read()
    head = '\n'.join(text.splitlines()[:40]) + '\n'
    feats, labels, total, discarded = D.load_libsvm_data(head, 5, 136)
    out = {
        'source': 'tensorflow_ranking/examples/data/train_numerical_elwc.tfrecord (first 3 of %d records); '
                  'train.txt (first 40 lines)' % len(records),
        'n_records_in_file': len(records),
        'file_crc32c': D.crc32c(buf),
        'records_b64': [base64.b64encode(r).decode('ascii') for r in take],
        'decoded': decoded,
        'libsvm_head': head,
        'libsvm': {'list_size': 5, 'num_features': 136, 'labels': labels, 'total': total, 'discarded': discarded,
                   'nonzero': [[b, d, k, feats[b][d][k]] for b in range(len(feats)) for d in range(5)
                               for k in range(136) if feats[b][d][k] != 0.0]},
    }
    path = os.path.join(ROOT, 'tests', 'golden', 'elwc_golden.json')
    json.dump(out, open(path, 'w'))
    print('wrote', path, os.path.getsize(path), 'bytes; records in file:', len(records))


if __name__ == '__main__':
    main()
