"""Importable alias of the ``gpv-1_amd`` package directory (a hyphen cannot be imported).

``import gpv1_amd.hip`` resolves to ``gpv-1_amd/hip.py``; all sub-modules live there.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'gpv-1_amd'))
__version__ = '0.1.0'
