"""Every operator and one full step against the oracle on shapes that sit on the kernels' dispatch boundaries: 33..45
electrons per spin (LDS Gauss-Jordan, GEMM-chain propagator), M just above / at the limits of the fused kernels (104, 105,
128, 130, 200), unequal spins, real and complex trials, populations above and below the work-group-tiled GEMM threshold."""
import pytest

from tests.test_gpu_sizes import test_midsize_generic as run_shape

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,na,nb,nw,cplx", [
    (64, 20, 40, 37, 70, False), (100, 30, 45, 33, 66, True), (130, 16, 20, 20, 70, False), (48, 10, 33, 33, 8, False),
    (200, 12, 10, 9, 65, True), (128, 8, 32, 32, 64, False), (104, 9, 32, 31, 64, False), (105, 9, 26, 25, 64, True),
])
def test_operators_and_step_on_boundary_shapes(M, K, na, nb, nw, cplx):
    run_shape(M, K, na, nb, nw, cplx)
