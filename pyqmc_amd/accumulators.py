"""Parameter-gradient accumulators — SURVEY.md §8(f2).

Public names, argument meaning and returned keys are those of the reference's ``LinearTransform``
(``pyqmc/observables/accumulators.py:98-185``) and ``StochasticReconfiguration`` / ``PGradTransform``
(``pyqmc/observables/stochastic_reconfiguration.py:22-176``); the implementation is this package's own:

* ``LinearTransform`` keeps, per optimised key, the flat positions of the selected entries; (de)serialisation is fancy
  indexing with those positions.
* ``StochasticReconfiguration.avg`` gets all three moments from ONE product on the fp64 matrix cores (``pqa_gram`` ->
  ``k_gram_mfma``): with ``A = [dp | E | 1]`` (nconf, p+2) and ``B = w f dp`` (nconf, p),
  ``A^T B = [[dpidpj], [dpH], [dppsi]]``.  The per-walker derivatives come from ``wf.pgradient()`` (``k_pgrad_det``,
  ``k_pgrad_mo``, ``k_j3_pgrad``).
"""

import numpy as np

from . import _ffi


class LinearTransform:
    """Optimised subset of a parameter dictionary as one real vector.

    ``to_opt[k]``: boolean array shaped like ``parameters[k]``; keys without any selected entry are ignored.  Layout of
    the vector (as the reference, ``accumulators.py:134-150``): selected entries key after key in C order (real parts),
    followed by the imaginary parts of the selected entries of the complex keys."""

    def __init__(self, parameters, to_opt=None):
        if to_opt is None:
            to_opt = {k: np.ones(np.shape(v), dtype=bool) for k, v in parameters.items()}
        self.to_opt, self.shapes, self.dtypes, self._pos = {}, {}, {}, {}
        for k, m in to_opt.items():
            m = np.asarray(m, dtype=bool)
            if not m.any():
                continue
            v = np.asarray(parameters[k])
            self.to_opt[k], self.shapes[k], self.dtypes[k] = m, v.shape, v.dtype
            self._pos[k] = np.flatnonzero(m.ravel())
        self.slices = {k: int(np.prod(s)) for k, s in self.shapes.items()}
        self.complex = {k: np.issubdtype(d, np.complexfloating) for k, d in self.dtypes.items()}
        self.nimag = {k: (len(self._pos[k]) if self.complex[k] else 0) for k in self._pos}
        self.nparams = int(sum(len(p) for p in self._pos.values()))
        flags = [np.full(len(self._pos[k]), self.complex[k]) for k in self._pos]
        self.complex_inds = np.concatenate(flags) if any(self.nimag.values()) else np.zeros(0, dtype=bool)

    def _gather(self, arrays, lead):
        cols = [np.asarray(arrays[k]).reshape(lead + (-1,))[..., p] for k, p in self._pos.items()]
        return np.concatenate(cols, axis=-1) if cols else np.zeros(lead + (0,))

    def serialize_parameters(self, parameters):
        x = self._gather(parameters, ())
        return np.concatenate((x.real, x[self.complex_inds].imag)) if len(self.complex_inds) else np.real(x)

    def serialize_gradients(self, pgrad):
        """(nconf, nparams [+ imaginary tail]) matrix of d log Psi / d p; the tail columns are i times their real twins."""
        if not self._pos:
            return np.zeros(0)
        nconf = np.shape(next(iter(pgrad.values())))[0]
        g = self._gather(pgrad, (nconf,))
        return np.concatenate((g, 1j * g[:, self.complex_inds]), axis=1) if len(self.complex_inds) else g

    def deserialize(self, wf, parameters):
        """Vector -> {key: array}: selected entries from the vector, the rest from ``wf.parameters``."""
        out, re0, im0 = {}, 0, self.nparams
        for k, p in self._pos.items():
            flat = np.array(wf.parameters[k], dtype=self.dtypes[k]).ravel()
            new = np.asarray(parameters[re0 : re0 + len(p)]).real.astype(self.dtypes[k])
            if self.complex[k]:
                new = new + 1j * np.asarray(parameters[im0 : im0 + len(p)])
                im0 += len(p)
            flat[p] = new
            re0 += len(p)
            out[k] = flat.reshape(self.shapes[k])
        return out


def nodal_regularization(grad2, nodal_cutoff=1e-3):
    """Pathak-Wagner weight (``stochastic_reconfiguration.py:22-46``): with the node-distance estimate ``r = 1/grad2`` and
    ``x = r / cutoff^2``, walkers with ``x < 1`` are weighted ``x (9 - 15 x + 7 x^2)``, the others 1.  Returns (mask, f)."""
    x = 1.0 / (np.asarray(grad2, dtype=float) * nodal_cutoff**2)
    near = x < 1.0
    return near, np.where(near, x * (9.0 + x * (-15.0 + 7.0 * x)), 1.0)


def device_gram(wf):
    """``(A, B) -> A^T B`` on the device that holds ``wf`` (``pqa_gram``).  Raises when ``wf`` has no device handle: the
    moment matrices of the product path are never formed on the host."""
    dev = next((d for d in (getattr(w, "_dev", None) for w in [wf, *getattr(wf, "wf_factors", ())]) if hasattr(d, "call")), None)
    if dev is None:
        raise RuntimeError("StochasticReconfiguration needs a wave function resident on the HIP device (pqa_gram)")

    def real_gram(a, b):
        a, b = _ffi.f64(a), _ffi.f64(b)
        c = np.empty((a.shape[1], b.shape[1]))
        dev.call("pqa_gram", a.shape[0], a.shape[1], b.shape[1], _ffi.ptr(a), _ffi.ptr(b), _ffi.ptr(c))
        return c

    def gram(a, b):
        if not (np.iscomplexobj(a) or np.iscomplexobj(b)):
            return real_gram(a, b)
        ar, ai, br, bi = np.real(a), np.imag(a), np.real(b), np.imag(b)  # (ar + i ai)^T (br + i bi), no conjugation
        return real_gram(ar, br) - real_gram(ai, bi) + 1j * (real_gram(ar, bi) + real_gram(ai, br))

    return gram


class StochasticReconfiguration:
    """Energy plus the moments ``dpH = <E f dp>``, ``dppsi = <f dp>``, ``dpidpj = <dp (f dp)^T>`` of the logarithmic
    parameter derivatives (``stochastic_reconfiguration.py:49-118``) and the SR step from their averages (:120-176).

    ``gram``: callable ``(A, B) -> A^T B``; default: the device product of the wave function's own handle (tests of the
    host logic inject a NumPy one)."""

    def __init__(self, enacc, transform, nodal_cutoff=1e-3, eps=1e-1, inverse_strategy="pseudo_inverse", verbose=False, gram=None):
        self.enacc, self.transform = enacc, transform
        self.nodal_cutoff, self.eps, self.inverse_strategy, self.verbose = nodal_cutoff, eps, inverse_strategy, verbose
        self._gram = gram

    def _derivatives(self, configs, wf, cutoff):
        dp = self.transform.serialize_gradients(wf.pgradient())
        en = self.enacc(configs, wf)
        return dp, en, dp * nodal_regularization(en["grad2"], cutoff)[1][:, None]

    def __call__(self, configs, wf):
        dp, d, fdp = self._derivatives(configs, wf, self.nodal_cutoff)
        d["dpH"] = d["total"][:, None] * fdp
        d["dppsi"] = fdp
        d["dpidpj"] = dp[:, :, None] * fdp[:, None, :]
        return d

    def avg(self, configs, wf, weights=None):
        dp, en, fdp = self._derivatives(configs, wf, 1e-3)  # the reference regularises with the default cut-off here (:105)
        nconf = configs.configs.shape[0]
        w = np.full(nconf, 1.0 / nconf) if weights is None else np.asarray(weights, dtype=float) / np.sum(weights)
        d = {k: np.tensordot(w, v, axes=(0, 0)) for k, v in en.items()}
        p = self.transform.nparams
        if p > 0:
            gram = self._gram or device_gram(wf)
            lhs = np.concatenate((dp, en["total"][:, None], np.ones((nconf, 1))), axis=1)
            m = gram(lhs, w[:, None] * fdp)
            d["dpidpj"], d["dpH"], d["dppsi"] = m[:-2], m[-2], m[-1]
        return d

    def keys(self):
        return self.enacc.keys().union(["dpH", "dppsi", "dpidpj"])

    def shapes(self):
        p = self.transform.nparams
        return {**self.enacc.shapes(), "dpH": (p,), "dppsi": (p,), "dpidpj": (p, p)}

    def delta_p(self, steps, data, verbose=False):
        """``[-step S^-1 g for step in steps]`` with ``g = 2 Re(<E dp> - <E><dp>)`` and the covariance
        ``S = Re(<dp dp^T> - <dp><dp>^T)`` of averaged ``data``; ``S^-1`` by truncated pseudo-inverse (``rcond = eps``)
        or as ``(S + eps 1)^-1``."""
        mean_dp = np.asarray(data["dppsi"])
        g = 2.0 * np.real(np.asarray(data["dpH"]) - data["total"] * mean_dp)
        S = np.real(np.asarray(data["dpidpj"]) - np.outer(mean_dp, mean_dp))
        if self.inverse_strategy == "pseudo_inverse":
            v = np.linalg.pinv(S, rcond=self.eps) @ g
        elif self.inverse_strategy == "regularized_inverse":
            v = np.linalg.solve(S + self.eps * np.identity(len(g)), g)
        else:
            raise ValueError("Invalid inverse strategy. Valid options are pseudo_inverse and regularized_inverse.")
        gn, vn = np.linalg.norm(g), np.linalg.norm(v)
        report = {"pgrad": gn, "SRdot": float(g @ v) / (vn * gn)}
        if verbose or self.verbose:
            print("Gradient norm: ", report["pgrad"], " SR dot: ", report["SRdot"])
        return [-s * v for s in steps], report


PGradTransform = StochasticReconfiguration
