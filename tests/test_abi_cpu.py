"""CPU-only: libsegnb_hip.so loads (no GPU needed: nothing is launched) and exports every entry point that
include/segnb_hip.h declares; the ctypes binding table (segnb._native) and the CPU restatement of the ABI
(oracle/abi_emulator.py) cover exactly the same set of names."""
import ctypes
import os
import re

import pytest

from oracle import abi_emulator
from segnb import _native as nv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, 'include', 'segnb_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)               # prose mentions functions too
    return set(re.findall(r'\b(segnb_\w+)\s*\(', hdr))


def test_header_declares_the_entry_points():
    names = _declared()
    assert len(names) >= 35
    for must in ('segnb_conv_fprop', 'segnb_conv_wgrad', 'segnb_bn_act_fwd', 'segnb_bn_bwd_apply_direct',
                 'segnb_seg_loss_reduce', 'segnb_sgd_step', 'segnb_tiles_merge', 'segnb_tune'):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(nv.LIB_PATH):
        pytest.skip('libsegnb_hip.so not built (run __graft_entry__.build())')
    lib = ctypes.CDLL(nv.LIB_PATH)
    missing = [n for n in sorted(_declared()) if not hasattr(lib, n)]
    assert not missing, 'declared in include/segnb_hip.h but not exported: %s' % missing


def test_binding_table_and_emulator_match_the_header():
    names = _declared()
    bound = set(nv.SIGNATURES) | set(nv.PLAIN)
    assert bound == names, (sorted(names - bound), sorted(bound - names))
    emu = abi_emulator.AbiEmulator()
    host_only = {'segnb_last_error', 'segnb_version', 'segnb_device_cus',      # no arithmetic to restate
                 'segnb_plan_begin', 'segnb_plan_end', 'segnb_plan_run', 'segnb_plan_destroy'}      # launcher plumbing
    missing = [n for n in sorted(names - host_only) if not hasattr(emu, n)]
    assert not missing, 'no CPU restatement for: %s' % missing


def test_emulator_switch_is_refused_outside_the_test_harness(monkeypatch):
    """the product must not be able to route ABI calls away from libsegnb_hip.so."""
    from segnb import _native as nv
    monkeypatch.delenv('SEGNB_TEST_HARNESS')
    with pytest.raises(RuntimeError):
        nv.set_backend_for_testing(object())
    nv.set_backend_for_testing(None)           # clearing is always allowed
    assert nv._test_backend is None
