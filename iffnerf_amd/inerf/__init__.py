"""Mirror of the reference's ``inerf`` package for the refinement stage ``pose_estimation/test.py:196-211`` enables with
``inerf_refinement=True``: the optimisation loop lives here, the march and its ray gradients in HIP."""
