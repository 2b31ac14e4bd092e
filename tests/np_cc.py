"""numpy restatement of ONE closed-shell CCSD update (src/ccsd.f90:1040-1312 intermediates, :1538-1732 amplitudes) for sizes the
loop-form oracle (oracle/afesp_oracle.c) cannot reach in test time: every o^3 v^3 sum is one dgemm over reshaped operands, and the
O(o^2 v^4) particle-particle ladder is evaluated for SAMPLED column pairs (a, b) only.  Test infrastructure, like oracle/: pinned to
the loop form by tests/test_oracle_golden.py::test_numpy_restatement_equals_the_loop_form (which the reference's bundled outputs pin),
used by tools/big_system_check.py on tensors downloaded from the device.  Index order = the reference's (arrays indexed [i, j, a, b]
etc.); formulas follow oracle/afesp_oracle.c:221-365 line by line."""
import numpy as np


def _mm(A, ia, B, ib, out_axes):
    """sum over the axes named alike: A, B with axis-name strings ia, ib -> array with axes out_axes (one dgemm)."""
    k = [c for c in ia if c in ib and c not in out_axes]
    fa = [c for c in ia if c not in k]
    fb = [c for c in ib if c not in k]
    At = np.transpose(A, [ia.index(c) for c in fa + k]).reshape(int(np.prod([A.shape[ia.index(c)] for c in fa])), -1)
    Bt = np.transpose(B, [ib.index(c) for c in k + fb]).reshape(-1, int(np.prod([B.shape[ib.index(c)] for c in fb])))
    C = (At @ Bt).reshape([A.shape[ia.index(c)] for c in fa] + [B.shape[ib.index(c)] for c in fb])
    names = fa + fb
    return np.transpose(C, [names.index(c) for c in out_axes])


def intermediates(t1, t2, oovv, ovov, vvov, oovo, oooo):
    """All intermediates of update_restricted_intermediates but the o v^3 tensor I_vovv_p (see vovv_p_cols)."""
    asym = 2.0 * t2 - t2.transpose(1, 0, 2, 3)                                       # :1063-1064
    c = t2 + np.einsum("ia,jb->ijab", t1, t1)                                        # :1071-1079
    w = 2.0 * oovv - oovv.transpose(0, 1, 3, 2)
    I = {"asym_t2": asym, "c_oovv": c}
    I["I_vo"] = _mm(w, "miea", t1, "me", "ai")                                       # :1085-1092
    wv = 2.0 * vvov - vvov.transpose(1, 0, 2, 3)                                     # 2<eb|ma> - <be|ma>
    I["I_vv"] = _mm(wv, "ebma", t1, "me", "ba") - _mm(w, "mneb", c, "mnea", "ba")    # :1096-1113
    del wv
    wo = 2.0 * oovo - oovo.transpose(1, 0, 2, 3)
    I["I_oo_p"] = _mm(wo, "miej", t1, "me", "ji") + _mm(oovv, "mief", asym, "mjef", "ji")   # :1115-1132
    I["I_oo"] = I["I_oo_p"] + _mm(t1, "je", I["I_vo"], "ei", "ji")                   # :1134-1137
    I["I_oooo"] = (oooo + _mm(c, "klef", oovv, "ijef", "klij") + _mm(t1, "ke", oovo, "ilej", "klij")
                   + _mm(t1, "le", oovo, "jkei", "klij"))                            # :1139-1156
    I["I_ovov"] = (ovov - 0.5 * _mm(oovv, "mibe", c, "mjae", "jbia") - _mm(oovo, "mibj", t1, "ma", "jbia")
                   + _mm(t1, "je", vvov, "ebia", "jbia"))                            # :1158-1191
    I["x_voov"] = _mm(vvov, "beia", t1, "je", "bjia")                                # :1275-1290
    I["I_voov"] = (oovv.transpose(3, 0, 1, 2) + _mm(oovv - 0.5 * oovv.transpose(0, 1, 3, 2), "imbe", t2, "mjea", "bjia")
                   - 0.5 * _mm(oovv, "imbe", c, "mjae", "bjia") + I["x_voov"] - _mm(oovo, "imbj", t1, "ma", "bjia"))   # :1193-1252
    I["I_ooov_p"] = (oovo.transpose(1, 0, 3, 2) + _mm(t2, "jkef", vvov, "efia", "jkia")
                     + _mm(t1, "je", I["x_voov"], "ekia", "jkia"))                   # :1302-1308
    return I


def vovv_p_cols(t1, oovv, ovov, vvov, a, b):
    """I_vovv_p(c, i, a, b) for one column pair (a, b): [v, o]   (:1255-1272, :1296-1299)"""
    return vvov[b, a, :, :].T - np.einsum("mic,m->ci", oovv[:, :, :, b], t1[:, a]) - np.einsum("mic,m->ci", ovov[:, a, :, :], t1[:, b])


def r1(t1, I, oovv, ovov, vvov, oovo):
    """T1 residual, Eq. 43 (:1569-1631)"""
    asym = I["asym_t2"]
    return (t1 @ I["I_vv"] - I["I_oo_p"] @ t1 + _mm(asym, "miea", I["I_vo"], "em", "ia")
            + _mm(2.0 * oovv - ovov.transpose(0, 2, 3, 1), "miea", t1, "me", "ia") - _mm(oovo, "mien", asym, "mnea", "ia")
            + _mm(asym, "mief", vvov, "efma", "ia"))


def r2_cols(t1, t2, I, oovv, ovov, vvov, vvvv_ab, a, b):
    """T2 residual before P(ia/jb), Eq. 44 (:1637-1716), for one column pair: r2[:, :, a, b] as [o, o]; vvvv_ab = <ef|ab> as [v, v]."""
    c, asym = I["c_oovv"], I["asym_t2"]
    x = t2[:, :, a, :] @ I["I_vv"][:, b]                                             # :1647
    x = x - np.einsum("mi,jm->ij", t2[:, :, b, a], I["I_oo"])                        # :1654-1664
    lad = np.tensordot(c, vvvv_ab, axes=([2, 3], [0, 1])) + np.tensordot(I["I_oooo"], c[:, :, a, b], axes=([2, 3], [0, 1]))
    x = x + 0.5 * lad                                                                # :1669, :1673
    x = x - np.einsum("mje,iem->ij", t2[:, :, a, :], I["I_ovov"][:, :, :, b])        # :1680-1695
    x = x - np.einsum("iem,mje->ij", I["I_ovov"][:, :, :, a], t2[:, :, :, b])
    x = x + np.einsum("mie,ejm->ij", asym[:, :, :, a], I["I_voov"][:, :, :, b])
    x = x + t1 @ vovv_p_cols(t1, oovv, ovov, vvov, a, b)                             # :1700   (i,e) x (e,j)
    x = x - np.einsum("m,ijm->ij", t1[:, a], I["I_ooov_p"][:, :, :, b])              # :1705-1715
    return x


def new_t2_cols(r2_ab, r2_ba, oovv, D2, a, b):
    """P(ia/jb), + v_oovv, Jacobi divide (:1720-1728) for one column pair"""
    return (r2_ab + r2_ba.T + oovv[:, :, a, b]) / D2[:, :, a, b]
