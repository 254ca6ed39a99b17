"""CPU oracle for the sampling-MPC control step of LeCAR-Lab/CoVO-MPC (quadjax).

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, link or execute anything under ``oracle/``; the product package
``covo_mpc_amd`` never does (and fails loudly without its HIP library).

PARITY UNPINNED.  The reference is Python-on-JAX; ``jax``/``jaxlib``/``flax``/
``chex``/``gymnax`` are not installed in the build container, there is no
network, and the reference ships no tests, fixtures or golden vectors
(SURVEY.md section 8c).  It therefore cannot be imported, executed or compiled
here, and nothing captured from it exists.  This oracle is a line-by-line
restatement of the reference's arithmetic (every function cites the reference
file:line it follows) pinned only by hand-derived known-answer values
(``tests/golden/kat.json``).  Third-party arithmetic
the reference delegates to unpinned ``jax`` (``setup.py:21``) -- the threefry
PRNG bitstream, ``jnp.linalg.eigh``/``cholesky``, ``jax.jacfwd`` -- is
restated from its published definition:
  * noise epsilon is an EXPLICIT input of every parity interface,
  * ``multivariate_normal`` = mean + chol_lower(cov) @ eps  (jax default
    ``method='cholesky'``),
  * eigh/cholesky = LAPACK via numpy (fp64),
  * jacfwd(jacfwd(f)) = exact Hessian (torch.func forward-over-forward on the
    fp64 torch restatement in ``ref_torch.py``; hyper-dual numbers in
    ``ref_np.hessian_hyperdual``).

Modules
  ref_np.py      numpy restatement (fp32 or fp64), vectorised over samples
  ref_torch.py   torch fp64 restatement of the Hessian objective (AD oracle)
  rng_np.py      Philox4x32-10 counter RNG + Box-Muller (build-defined stream)
  covo_oracle.c  plain-C restatement of the hot loops (fp32 + fp64), also the
                 timed "port" CPU baseline of bench.py
"""
