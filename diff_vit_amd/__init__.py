"""Importable alias of the ``diff-vit_amd/`` package directory (a hyphen is not a valid module name).

``import diff_vit_amd`` executes ``diff-vit_amd/__init__.py`` with this module as the package, so
``diff_vit_amd.plan`` etc. resolve to the files under ``diff-vit_amd/``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'diff-vit_amd')
__path__[:] = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f
