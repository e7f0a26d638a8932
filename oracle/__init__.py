"""CPU oracle for the TF-Ranking loss-and-score hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ranking_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.

What it is: an op-for-op torch-CPU (fp32) restatement of the reference's
TensorFlow op graph for SURVEY.md section 8(a).  It materialises the same
``[B, L, L]`` tensors the reference does and relies on torch autograd for the
backward pass, exactly as the reference relies on TF autodiff.

Parity pinning: the reference cannot be imported here (TensorFlow is absent,
no network) and has no compiled sources, so the oracle is pinned against the
known-answer literals of the reference's own unit tests (transcribed, with
file:line citations, in ``tests/test_oracle_golden.py`` and
``tests/golden/*.json``).  Unpinned, by construction of the reference:
  * tie order (the reference shuffles ties randomly, utils.py:100-112) --
    the oracle uses the stable "lower index first" rule that tf.math.top_k
    follows when shuffling is off (utils_test.py:108-109);
  * the Gumbel noise stream (TF Philox) -- noise is injected;
  * gradients (the reference has no gradient tests) -- autograd + fp64.
"""
from . import tfr_ref  # noqa: F401
