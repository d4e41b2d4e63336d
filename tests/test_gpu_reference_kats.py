"""The reference's own known-answer tests, run against the HIP path through the C ABI."""
import pytest

from tests import kats
from tests.backends import GpuBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend(ctx):
    return GpuBackend(ctx)


@pytest.mark.parametrize("kat", kats.NORMALS_KATS + kats.ICP_KATS + kats.P2PL_KATS + kats.GICP_KATS + kats.KISS_KATS, ids=lambda f: f.__name__)
def test_reference_kat(backend, kat):
    kat(backend)


@pytest.mark.parametrize("kat", kats.NORMALS_RADIUS_KATS, ids=lambda f: f.__name__)
def test_reference_kat_radius(backend, kat):
    kat(backend)
