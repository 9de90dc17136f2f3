"""sage_amd — MI355X-native engine for lazear/sage's fragment-index search-and-score path.

The package holds only what that path needs: csrc/ (HIP kernels, the C ABI of include/sage_hip.h, the host
index builder) and a thin host-side mirror of the reference's Scorer / IndexedDatabase interface (api.py).
"""
from .api import (DatabaseParameters, DeviceBatch, DeviceDatabase, IndexedDatabase, ProcessedSpectrum, RawSpectrum,
                  Scorer, ScorerParams, SpectrumBatch, SpectrumProcessor, Tolerance, device_count, predict_rt, rescore)

__all__ = ["DatabaseParameters", "DeviceBatch", "DeviceDatabase", "IndexedDatabase", "ProcessedSpectrum",
           "RawSpectrum", "Scorer", "ScorerParams", "SpectrumBatch", "SpectrumProcessor", "Tolerance", "device_count",
           "predict_rt", "rescore"]
