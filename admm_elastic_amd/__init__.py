"""Import shim: the product package lives in the directory ``admm-elastic_amd/`` (the name the
build contract asks for), which is not a valid Python identifier.  This shim makes it importable as
``admm_elastic_amd`` by pointing the package search path at that directory."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "admm-elastic_amd"))

from ._impl import *  # noqa: F401,F403,E402
from ._impl import __all__  # noqa: E402
