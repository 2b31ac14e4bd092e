"""Config 5 (o=20, v=200, the size BASELINE.json's roofline numbers are quoted on): the oracle cannot run a whole step at
this size in test time, so parity is shown through size-independent properties and through oracle evaluations of a few
triples on the tensors the device holds."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
O, V = 20, 200


@pytest.fixture(scope="module")
def big():
    from afesp_amd.capi import Engine
    e = Engine(0)
    e.synthetic_init(O, V, 0.005, 12345, 8)
    e.ccsd_energy()
    energies = []
    for _ in range(2):
        energies.append(e.ccsd_iterate()[0])
        e.ccsd_diis()
    e.fullsize_energies = energies
    yield e
    e.close()


def test_amplitude_symmetry_and_energy_functional(big):
    """t2(i,j,a,b) = t2(j,i,b,a) is preserved by the equations (and by the a<=b evaluation of the pp-ladder); the energy the
    device reports is the functional of ccsd.f90:1775 evaluated on the host from the downloaded tensors."""
    t1, t2 = big.amplitudes()
    assert np.max(np.abs(t2 - t2.transpose(1, 0, 3, 2))) < 1e-13
    v_oovv = big.tensor("v_oovv")
    big.ccsd_energy()
    e, rms, conv = big.ccsd_energy()
    w = 2.0 * v_oovv - v_oovv.transpose(0, 1, 3, 2)
    ref = float(np.sum(w * (t2 + np.einsum("ia,jb->ijab", t1, t1))))
    assert abs(e - ref) < 1e-10 * max(1.0, abs(ref))
    assert rms == 0.0           # second call in a row: t2_old == t2 (ccsd.f90:1803-1806)


def test_triples_shards_add_up_and_match_oracle_on_device_tensors(big):
    nt = big.ntriples()
    assert nt == O * (O + 1) * (O + 2) // 6
    full = big.do_ccsd_t_spatial()
    plain = big.do_ccsd_t_spatial_plain()          # the variant bench.py times: E[T], E(T) with one Z evaluation per element
    assert np.max(np.abs(plain - full[:2])) < 1e-12 * np.max(np.abs(full[:2]))
    cuts = [0, 2, nt // 7, nt // 2, nt - 5, nt]
    parts = sum(big.do_ccsd_t_spatial(a, b) for a, b in zip(cuts[:-1], cuts[1:]))
    assert np.max(np.abs(parts - full)) < 1e-11 * np.max(np.abs(full))
    # first two sorted triples (0,0,0), (0,0,1) = ordered triples 0, 1, o, o^2 of the reference's enumeration
    t1, t2 = big.amplitudes()
    f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
    e = np.concatenate([-2.0 + np.arange(O) / (O - 1), 1.0 + 2.0 * np.arange(V) / (V - 1)])
    vvov, oovo, oovv = f(big.tensor("v_vvov")), f(big.tensor("v_oovo")), f(big.tensor("v_oovv"))
    L = orc.lib()
    ref = np.zeros(4)
    for lo, hi in ((0, 2), (O, O + 1), (O * O, O * O + 1)):
        out = np.zeros(4)
        L.orc_ccsd_t(O, V, e, f(t1), f(t2), vvov, oovo, oovv, lo, hi, out)
        ref += out
    got = big.do_ccsd_t_spatial(0, 2)
    assert np.max(np.abs(got - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))
