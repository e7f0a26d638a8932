"""Achieved error vs tolerance of every parity check (see tests/conftest.py: printed at the end of the session)."""
import os

MARGINS = {}


def record_margin(what, err, bar):
    test = os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]
    key = what or test
    frac = float(err) / float(bar) if bar else 0.0
    cur = MARGINS.get(key)
    if cur is None or frac > cur['used']:
        MARGINS[key] = {'max_err': float(err), 'bar': float(bar), 'used': frac, 'test': test}
