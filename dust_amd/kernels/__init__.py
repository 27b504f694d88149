"""Kernel selectors with the reference's names (dust/kernels/base_kernels.py:39, composite_kernels.py:33) plus a
`RBFKernel` tag with gpytorch-RBF semantics and the new `IMQ`.  They carry parameters only; the Gram / phi math is in
csrc/stein.hpp and csrc/bandwidth.hpp."""


class RBF:
    def __init__(self, bandwidth=-1, bw_scale=1.0, analytic_grad=True, minimum_bw=1e-5, **kwargs):
        self.ell, self.ell_scale, self.analytic_grad, self.minimum_bw = bandwidth, bw_scale, analytic_grad, minimum_bw


class iid_mp:
    def __init__(self, base_kernel=None, ctrl_dim=1, indep_controls=True, **kwargs):
        self.base_kernel = base_kernel if base_kernel is not None else RBF()
        self.ctrl_dim, self.indep_controls = ctrl_dim, indep_controls


class RBFKernel:
    """`kernel: rbf` in the demo yamls = gpytorch.kernels.RBFKernel(); lengthscale stays softplus(0) = ln 2 (svmpc.py:78)."""


class IMQ:
    """k(x, y) = (1 + |x - y|^2 / ell^2)^(-1/2); new (BASELINE.json), no reference implementation."""

    def __init__(self, ell=1.0):
        self.ell = float(ell)


def kernel_config(kernel):
    if kernel is None or isinstance(kernel, RBFKernel) or type(kernel).__name__ == "RBFKernel":
        return dict(kernel="K1")
    if isinstance(kernel, iid_mp):
        return dict(kernel="K2" if kernel.indep_controls else "K2shared", bw_scale=kernel.base_kernel.ell_scale,
                    k2_bandwidth=kernel.base_kernel.ell, k2_minimum_bw=kernel.base_kernel.minimum_bw)
    if isinstance(kernel, IMQ):
        return dict(kernel="IMQ", imq_ell=kernel.ell)
    if isinstance(kernel, RBF):
        raise ValueError("a bare RBF kernel breaks SVMPC.phi's broadcast in the reference too (svmpc.py:71); wrap it in iid_mp")
    raise NotImplementedError("kernel %r has no HIP implementation" % (kernel,))
