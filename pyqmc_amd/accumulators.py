"""Parameter-gradient accumulators over the wave-function protocol — SURVEY.md §8(f2).

Counterparts of ``LinearTransform`` (``pyqmc/observables/accumulators.py:98-185``) and
``StochasticReconfiguration`` / ``PGradTransform`` (``pyqmc/observables/stochastic_reconfiguration.py:22-176``).
The per-walker parameter derivatives come from ``wf.pgradient()`` (HIP kernels ``k_pgrad_det`` / ``k_pgrad_mo`` /
``k_j3_pgrad`` and the resident Jastrow sums); what is here is the reference's own host logic on top of them: the
parameter (de)serialisation, the nodal regularisation, the three moments ``dpH``, ``dppsi``, ``dpidpj`` and the
SR step.  Same names, argument meaning and return keys as the reference.
"""

import numpy as np


class LinearTransform:
    """Linearise a dictionary of wave-function parameters (``accumulators.py:98-185``).

    ``to_opt[k]`` is a boolean array of the shape of ``parameters[k]`` selecting what is optimised; keys whose mask is
    all False are dropped.  Complex parameters contribute their real parts first and their imaginary parts at the end of
    the serialised vector, as in the reference."""

    def __init__(self, parameters, to_opt=None):
        parameters = {k: np.asarray(v) for k, v in parameters.items()}
        if to_opt is None:
            to_opt = {k: np.ones(p.shape, dtype=bool) for k, p in parameters.items()}
        self.to_opt = {k: np.asarray(o, dtype=bool) for k, o in to_opt.items() if np.any(o)}
        self.shapes = {k: parameters[k].shape for k in self.to_opt}
        self.slices = {k: int(np.prod(s)) for k, s in self.shapes.items()}
        self.dtypes = {k: parameters[k].dtype for k in self.to_opt}
        self.complex = {k: d == complex for k, d in self.dtypes.items()}
        self.nimag = {k: int(self.to_opt[k].sum()) if c else 0 for k, c in self.complex.items()}
        if any(self.nimag.values()):
            self.complex_inds = np.concatenate([np.full(int(self.to_opt[k].sum()), c, dtype=bool) for k, c in self.complex.items()])
        else:
            self.complex_inds = np.asarray([], dtype=bool)
        self.nparams = int(np.sum([v.sum() for v in self.to_opt.values()]))

    def serialize_parameters(self, parameters):
        if len(self.to_opt) == 0:
            return np.zeros(0)
        params = np.concatenate([np.asarray(parameters[k])[opt] for k, opt in self.to_opt.items()])
        return np.concatenate((params.real, params[self.complex_inds].imag))

    def serialize_gradients(self, pgrad):
        """(nconf, nparams) derivative matrix; frozen entries are dropped."""
        grads = [np.asarray(pgrad[k])[:, opt] for k, opt in self.to_opt.items()]  # C-order of the masked entries, as compress_cols
        if len(grads) == 0:
            return np.zeros(0)
        grads = np.concatenate(grads, axis=1)
        return np.concatenate((grads, grads[:, self.complex_inds] * 1j), axis=1)

    def deserialize(self, wf, parameters):
        """Serialised vector -> parameter dictionary (frozen entries taken from ``wf.parameters``)."""
        n, m, d = 0, self.nparams, {}
        for k, opt in self.to_opt.items():
            opt_ = opt.flatten()
            n_p = int(np.sum(opt_))
            flat = np.zeros(self.slices[k], dtype=self.dtypes[k])
            flat[~opt_] = np.asarray(wf.parameters[k])[~opt]
            flat[opt_] = np.real(parameters[n : n + n_p])
            if self.complex[k]:
                m_p = self.nimag[k]
                flat[opt_] += parameters[m : m + m_p] * 1j
                m += m_p
            d[k] = flat.reshape(self.shapes[k])
            n += n_p
        return d


def nodal_regularization(grad2, nodal_cutoff=1e-3):
    """Pathak-Wagner regularisation (``stochastic_reconfiguration.py:22-46``): walkers closer to the node than
    ``nodal_cutoff`` (distance estimate r = 1/|grad log psi|^2) get the polynomial weight 9x - 15x^2 + 7x^3,
    x = r/cutoff^2; returns (mask, f)."""
    r = 1.0 / grad2
    mask = r < nodal_cutoff**2
    c = 7.0 / nodal_cutoff**6
    b = -15.0 / nodal_cutoff**4
    a = 9.0 / nodal_cutoff**2
    f = a * r + b * r**2 + c * r**3
    f[np.logical_not(mask)] = 1.0
    return mask, f


class StochasticReconfiguration:
    """Energy accumulator plus the moments of the logarithmic parameter derivatives, and the SR step computed from their
    averages (``stochastic_reconfiguration.py:49-176``)."""

    def __init__(self, enacc, transform, nodal_cutoff=1e-3, eps=1e-1, inverse_strategy="pseudo_inverse", verbose=False):
        self.enacc = enacc
        self.transform = transform
        self.nodal_cutoff = nodal_cutoff
        self.eps = eps
        self.inverse_strategy = inverse_strategy
        self.verbose = verbose

    def __call__(self, configs, wf):
        pgrad = wf.pgradient()
        d = self.enacc(configs, wf)
        energy = d["total"]
        dp = self.transform.serialize_gradients(pgrad)
        _, f = nodal_regularization(d["grad2"], self.nodal_cutoff)
        dp_regularized = dp * f[:, np.newaxis]
        d["dpH"] = energy[:, np.newaxis] * dp_regularized
        d["dppsi"] = dp_regularized
        d["dpidpj"] = np.einsum("ij,ik->ijk", dp, dp_regularized)
        return d

    def avg(self, configs, wf, weights=None):
        nconf = configs.configs.shape[0]
        weights = np.ones(nconf) if weights is None else weights
        weights = weights / np.sum(weights)
        pgrad = wf.pgradient()
        den = self.enacc(configs, wf)
        energy = den["total"]
        dp = self.transform.serialize_gradients(pgrad)
        _, f = nodal_regularization(den["grad2"])  # the reference uses the default cutoff here (:105)
        dp_regularized = dp * f[:, np.newaxis]
        d = {k: np.average(it, weights=weights, axis=0) for k, it in den.items()}
        if self.transform.nparams > 0:
            wdp = weights[:, np.newaxis] * dp_regularized
            d["dpH"] = energy @ wdp
            d["dppsi"] = np.average(dp_regularized, weights=weights, axis=0)
            d["dpidpj"] = dp.T @ wdp  # the (nparams x nconf)(nconf x nparams) GEMM
        return d

    def keys(self):
        return self.enacc.keys().union(["dpH", "dppsi", "dpidpj"])

    def shapes(self):
        n = self.transform.nparams
        d = {"dpH": (n,), "dppsi": (n,), "dpidpj": (n, n)}
        d.update(self.enacc.shapes())
        return d

    def delta_p(self, steps, data, verbose=False):
        """Parameter changes ``-step * S^-1 g`` for every step length, from averaged data (keys as ``keys()``)."""
        pgrad = 2 * np.real(data["dpH"] - data["total"] * data["dppsi"])
        Sij = np.real(data["dpidpj"] - np.einsum("i,j->ij", data["dppsi"], data["dppsi"]))
        if self.inverse_strategy == "pseudo_inverse":
            invSij = np.linalg.pinv(Sij, rcond=self.eps)
        elif self.inverse_strategy == "regularized_inverse":
            invSij = np.linalg.inv(Sij + self.eps * np.eye(Sij.shape[0]))
        else:
            raise ValueError("Invalid inverse strategy. Valid options are pseudo_inverse and regularized_inverse.")
        v = invSij @ pgrad
        dp = [-step * v for step in steps]
        report = {"pgrad": np.linalg.norm(pgrad), "SRdot": np.dot(pgrad, v) / (np.linalg.norm(v) * np.linalg.norm(pgrad))}
        if verbose or self.verbose:
            print("Gradient norm: ", report["pgrad"], " SR dot: ", report["SRdot"])
        return dp, report


PGradTransform = StochasticReconfiguration
