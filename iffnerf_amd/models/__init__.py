"""Host-side mirror of the reference's ``models`` package for the hot path (same module and class names).

``iffnerf_amd.install()`` registers these modules under the reference's top-level names so that the
reference driver imports them unchanged.
"""
