"""pyqmc_amd — MI355X-native walker-batched trial-wave-function evaluator that drops in
behind PyQMC's wave-function protocol (see DESIGN.md / INTEGRATION.md).

Importing the package does not load the HIP library; the first object that needs the GPU
does, and raises if ``pyqmc_amd/lib/libpyqmc_amd.so`` is not built.
"""

from . import systems  # noqa: F401
from .configs import OpenConfigs, OpenElectron  # noqa: F401
from .dmc import branch, dmc_propagate, rundmc  # noqa: F401
from .energy import EnergyAccumulator  # noqa: F401
from .ecp_batched import ECPAccumulator  # noqa: F401
from .func3d import CutoffCuspFunction, PolyPadeFunction, default_jastrow_basis  # noqa: F401
from .systems import initial_guess  # noqa: F401
from .vmc import vmc, vmc_worker  # noqa: F401
from .wf import DeviceWF, JastrowSpin, MultiplyWF, Slater, ThreeBodyJastrow, generate_wf  # noqa: F401
from . import obdm  # noqa: F401
from .accumulators import LinearTransform, PGradTransform, StochasticReconfiguration  # noqa: F401
from .obdm import OBDMAccumulator  # noqa: F401
from .tbdm import TBDMAccumulator  # noqa: F401

__version__ = "0.1.0"
from . import chkfile, hdf5lite  # noqa: F401,E402  (PySCF checkpoint ingest without an HDF5 library)
