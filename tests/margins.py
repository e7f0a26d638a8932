"""Achieved error vs tolerance of every parity check (see tests/conftest.py: printed at the end of the session)."""
import os

MARGINS = {}


def record_margin(what, err, bar, pin=False):
    """pin=True: a BASELINE-configuration check -- its line is printed whatever its rank among the margins."""
    test = os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]
    key = what or test
    frac = float(err) / float(bar) if bar else 0.0
    cur = MARGINS.get(key)
    if cur is None or frac > cur['used']:
        MARGINS[key] = {'max_err': float(err), 'bar': float(bar), 'used': frac, 'test': test,
                        'pin': bool(pin) or bool((cur or {}).get('pin'))}
    elif pin:
        cur['pin'] = True
