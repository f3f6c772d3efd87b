"""Namespace-extending package: modules of the same package further down sys.path (the reference checkout) stay
importable; the sub-packages defined here win."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
