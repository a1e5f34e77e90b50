"""oracle/ -- CPU restatement of the reference algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import or execute anything in this directory, and there only as
the *checker*.  The product path (distributedconvrl-pde-control_amd/) never imports it
and fails loudly when the HIP library is missing.

Every function cites the reference file:line it restates (paths relative to the
reference repo janstenner/DistributedConvRL-PDE-Control).

Parity pinning:
  * KS CNAB2 step, KS prepare_action, KS reward_function      : PINNED by the reference's
    own logged trajectories (tests/golden/ks22_hook.npz, ks200_hook.npz,
    ks22_global_hook.npz) to ~1e-15 / 1e-14 / 1e-10.
  * Keller-Segel RHS + RK4, prepare_action, reward_function   : PINNED by
    tests/golden/kseg_hook.npz (reference used adaptive RK4 at tol 1e-8; fixed-step RK4
    with 32 sub-steps agrees to ~1e-8).
  * Fluid pseudo-spectral RHS / rk4                            : PARITY UNPINNED (the
    reference saved no fluid trajectory); pinned only by analytic known-answer tests.
  * MLP forward/backward, DDPG update, ADAM, Polyak           : PARITY UNPINNED by any
    reference artifact (saved weights do not reproduce saved actions: exploration noise);
    pinned by finite-difference gradient checks and an independent torch-autograd
    cross-check whose outputs are committed in tests/golden/nn_torch_golden.npz.
"""
