"""transmittances/{Erlang,Davis,DavisWeinstein}Transmittance.cpp of the reference (compiled from its own sources by oracle/Makefile.ref, driver
oracle/ref_driver/ref_kat4.cpp) against the oracle's restatement: the four forms, sigmaBar and both samplers, bit for bit."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "ref_kat4.json")))
MODELS = {"erlang": (5, [2.5, 0, 0]), "davis": (6, [1.6, 0, 0]), "davis_weinstein": (7, [0.75, 1.3, 0]), "davis_weinstein_h1": (7, [1.0, 0.7, 0])}


def taus():
    out = []
    for i in range(48):
        x = np.float32(0.0) if i == 0 else np.float32(0.01)*np.float32(i*i) + np.float32(1e-3)*np.float32(i)
        out.append([x, x*np.float32(0.5) + np.float32(0.125), x*np.float32(1.75)])
    return np.array(out, np.float32)


@pytest.mark.parametrize("case", KAT["transmittance"], ids=lambda c: c["name"])
def test_transmittance_model(oracle_py, case):
    t, param = MODELS[case["name"]]
    param = np.array(param, np.float32)
    tau = taus()
    forms = np.zeros((48, 4, 3), np.float32)
    sigma_bar = C.c_float()
    surf = np.zeros(192, np.float32)
    med = np.zeros(192, np.float32)
    L = oracle_py.lib()
    L.oracle_kat_transmittance.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p]
    L.oracle_kat_transmittance(t, param.ctypes.data, 0, tau.ctypes.data, 48, forms.ctypes.data, C.byref(sigma_bar), case["seed"], 192, surf.ctypes.data,
                               med.ctypes.data)
    assert np.float32(sigma_bar.value).view(np.uint32) == case["sigma_bar"]
    ref_forms = np.array(case["forms"], np.uint32).reshape(48, 4, 3)
    same = forms.view(np.uint32) == ref_forms
    both_nan = np.isnan(forms) & np.isnan(ref_forms.view(np.float32))
    assert (same | both_nan).all(), f"forms differ at {np.argwhere(~(same | both_nan))[:5].tolist()}"
    assert surf.view(np.uint32).tolist() == case["sample_surface"]
    assert med.view(np.uint32).tolist() == case["sample_medium"]
