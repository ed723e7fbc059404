"""Host-side mirror of the reference's ``dense_correspondence`` package, restricted to the training hot path
(SURVEY.md section 8): same module paths, class / function names, argument order and return values, with the
arithmetic executed by the gfx950 kernels in ``dcn_hip``."""
