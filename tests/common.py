"""Shared synthetic-input generator: lives in the package (ranking_amd/synthetic.py) so that
bench.py and smoke() do not depend on the test package; re-exported here for the tests."""
from ranking_amd.synthetic import make_batch, make_weights  # noqa: F401
