"""mydetection_amd -- MI355X-native single-stage detection inference path.

Host-side mirror of duanzhiihao/myDetection's plug-in surface (models.registry,
models.general, utils.structures, utils.bbox_ops, api.detection) over hand-written
gfx950 HIP kernels (csrc/, C ABI in include/mydet.h).  GPU only: no CPU fallback.
"""
import os

PROJECT_ROOT = os.path.dirname(os.path.abspath(__file__))     # plays settings.PROJECT_ROOT (settings.py:9)

__version__ = '0.1.0'
