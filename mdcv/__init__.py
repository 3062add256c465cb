"""Import alias for the package directory ``mit-driverless-cv-traininginfra_amd/`` (not a valid identifier)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mit-driverless-cv-traininginfra_amd")]
exec(open(_os.path.join(__path__[0], "__init__.py")).read())
