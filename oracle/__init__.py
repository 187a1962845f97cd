"""CPU oracle of the mx-DeepIM hot path — TEST INFRASTRUCTURE ONLY.

Plain numpy / C restatement of the reference algorithm, used as the parity checker by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing
under ``mx_deepim_amd/`` imports this package; the product path has no CPU fallback.

Pinning status:
  * S-group (RT_transform, calc_RT_delta, se3_mul/inverse, quat/euler algebra) and F2
    (calc_flow) are pinned against the reference's own Python, imported in the build
    container from /root/reference (tests/golden/make_golden.py → tests/golden/*.npz),
    plus the reference's doctest known-answers and the Transform3D 1e-4 check
    (deepim/operator_py/transform3d.py:394-410).
  * Z-group sampler, N-group conv/deconv/FC and F1 follow third-party MXNet 1.2 /
    the reference .cu, which cannot run here: PARITY UNPINNED by reference tests; they
    are cross-checked against torch-CPU (conv2d, conv_transpose2d, grid_sample
    align_corners=True) in tests/ only.

NumPy-1.x scalar promotion (the reference's era: MXNet 1.2 needs numpy < 1.15) is
restated with explicit dtypes, because this container runs NumPy 2 where
``np.float32(x) * 2.0`` stays float32.
"""
