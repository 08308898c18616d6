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
  * composed Model.__call__ / MLP.__call__ / train_step: the reference holds no
    test for them and flax/optax are not installable here, so they are pinned
    only by the published parameter counts (9,007,493 / 835,205 / 713,230 /
    615,740) and by being compositions of pinned leaves: PARITY UNPINNED for
    the composition itself.

Summation-order contract (bit-exact sample indices): `integrate_weights` and
the softmax denominator inside `invert_cdf` accumulate strictly left-to-right
in the working dtype, which is the order the HIP resampling kernel uses.
"""
