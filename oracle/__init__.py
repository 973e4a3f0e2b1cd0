"""CPU oracle for the DiTree expansion path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy f64 for the geometry, torch-CPU fp32 for
the denoiser) of the algorithm the reference implements in
``planners/RRT.py``, ``planners/base_planner.py``, ``car_env.py``,
``common/map_utils.py``, ``lidar_sim/lidar_2d_sim.py``, ``policies/fm_policy.py``
and ``model/diffusion/*``.  Every function cites the reference file:line it
follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product package
``ditreeonlineplanner_amd`` never imports anything from here.

Pinning status (see DESIGN.md "Oracle"):
  * collision, local map, lidar, sampler pre/post-processing, flow schedule,
    1-D U-Net, KD-tree nearest neighbour and the B = 1 planner loop are pinned by
    golden vectors produced in the build container by importing the reference's
    own modules (``tests/golden/make_golden.py``).
  * car dynamics (``car_env.py`` needs casadi + gymnasium, absent) and the
    ResNet-18-GN encoder (needs torchvision, absent) are restated from the text:
    **parity unpinned** for those two pieces beyond closed-form known answers.
"""
