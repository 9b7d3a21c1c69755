"""csrc/field_constants.inc (the Montgomery constants of the four fields the kernels compute in) is what
tools/gen_constants.py derives from the moduli alone, and those moduli are the oracle's -- Fp is the literal the reference
holds at src/chips/bitify.rs:461."""
import importlib.util
import os

import pasta as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "battlezips-halo2_amd")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_constants", os.path.join(PKG, "tools", "gen_constants.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_constants_are_the_generators_output():
    assert open(os.path.join(PKG, "csrc", "field_constants.inc")).read() == _gen().render()


def test_generator_moduli_are_the_oracles():
    fields = {name: p for name, _, p in _gen().FIELDS}
    assert fields["FpParams"] == O.FP.p == 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001   # src/chips/bitify.rs:461
    assert fields["FqParams"] == O.FQ.p
    assert fields["BnFrParams"] == O.BN_FR.p and fields["BnFqParams"] == O.BN_FQ.p
