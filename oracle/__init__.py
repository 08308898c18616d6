"""CPU oracle for the MultiNeRF per-ray hot path.  TEST INFRASTRUCTURE ONLY.

This package is a torch-CPU (fp32 by default, fp64 on request) restatement of
the reference's JAX math, one function per reference function, each citing the
reference file:line it follows.  It exists so the HIP kernels can be checked
for parity.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it; nothing under `multinerf_amd/` does, and the
product path raises if the HIP library is missing rather than fall back here.

Pinning status:
  * leaves (stepfun / render / coord / math / ref_utils / geopoly / image):
    pinned against (a) the golden vectors and known-answer tests held by the
    reference's own tests/ and (b) outputs of the reference's own source files
    executed in this container with a NumPy stand-in for `jax.numpy`
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * composed Model.__call__ / MLP.__call__ / loss terms / clip_gradients: the
    reference holds no test for them and flax / gin are not installable here;
    they are pinned against the reference's OWN internal/models.py and
    internal/train_utils.py executed in this container on small stand-ins for
    flax.linen / gin / jax.random with exact complex-step derivatives
    (tests/golden/make_golden_models.py -> tests/golden/models.npz, held at
    rtol 1e-9 in float64 by tests/test_oracle_models_golden.py), and by the
    published parameter counts.
  * PARITY UNPINNED: optax.adam (restated from its published algorithm) and
    jax's PRNG bit streams (every draw is an explicit input here).

Summation-order contract (bit-exact sample indices): the renormalisation sum of
`max_dilate_weights`, the softmax denominator and the CDF inside `invert_cdf`
accumulate in the BLOCKED order of the HIP level kernel (csrc/resample.hip: 16
lanes per ray; each lane its contiguous chunk of ceil(len/16) elements left to
right, then the 16 chunk sums left to right; `stepfun.blocked_cumsum`), in the
working dtype.
"""
