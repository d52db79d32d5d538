"""Preconditioned conjugate gradients with the call surface of the reference's `cg.py`.

``ConjugateGradients(A_apply_function, b, x0, tol, max_iter, early_stopping, M_inv_apply).solve()``
solves one system (``b`` of shape (n,)) or a batch with per-row early stopping (``b`` of shape (B, n));
``iters_completed`` is set as in the reference (cg.py:152, :243).

Two execution paths, identical semantics:

* when ``A_apply_function`` is one of this package's EFGP operators (`efgpnd.create_A_mean/A_var`)
  and the preconditioner is absent or this package's Jacobi object, the whole solve runs on the
  GPU inside ``efgp_cg_solve`` (fused HIP kernels + rocFFT; no per-iteration host work);
* for any other callable (or a dense matrix) the loop below runs with torch tensor ops on whatever
  device the tensors live on -- this is the generic compatibility path for user-defined operators,
  restating cg.py:86-153 (single) and :155-244 (batched).
"""
import torch


class ConjugateGradients:
    def __init__(self, A_apply_function, b, x0, tol=1e-6, max_iter=None, early_stopping=True, M_inv_apply=None):
        self.device = b.device
        if isinstance(A_apply_function, torch.Tensor):
            mat = A_apply_function
            self.A_apply_function = lambda v: mat @ v
            self.dtype = mat.dtype
        elif callable(A_apply_function):
            self.A_apply_function = A_apply_function
            self.dtype = x0.dtype
        else:
            raise ValueError("A_apply_function must be a torch.Tensor or a callable")
        self._operator = A_apply_function
        self.b = b.to(dtype=self.dtype, device=self.device)
        self.x0 = x0.to(dtype=self.dtype, device=self.device)
        self.is_batched = b.dim() > 1
        self.tol = tol
        self.div_eps = 1e-16
        if max_iter is not None:
            self.max_iter = max_iter
        else:
            self.max_iter = 2 * (self.b.shape[1] if self.is_batched else len(self.b))
        self.early_stopping = early_stopping
        self.M_inv_apply = M_inv_apply
        self.iters_completed = 0
        self.row_iters = None

    # ------------------------------------------------------------------------------------
    def solve(self):
        fused = self._fused_spec()
        if fused is not None:
            return self._solve_fused(*fused)
        return self._solve_batched() if self.is_batched else self._solve_single()

    def _fused_spec(self):
        """(operator, jacobi diagonal or None) when the solve can run inside efgp_cg_solve."""
        op = self._operator
        if not getattr(op, "_efgp_fusable", False) or not self.b.is_complex():
            return None
        pre = self.M_inv_apply
        if pre is None:
            return op, None
        if getattr(pre, "_efgp_jacobi", False):
            return op, pre.diag
        return None

    def _solve_fused(self, op, diag):
        from efgp_hip import cg_solve
        with torch.no_grad():
            x, iters, rows = cg_solve(op.toeplitz._op, op.ws, op.sigmasq, op.variant, self.b, self.x0, self.tol,
                                      max_iter=self.max_iter, early_stop=self.early_stopping, diag=diag,
                                      batched=self.is_batched)
        self.iters_completed = iters
        self.row_iters = rows
        return x.to(device=self.device, dtype=self.dtype)

    # ------------------------------------------------------------------------------------
    def _precond(self, r, clone):
        if self.M_inv_apply is not None:
            return self.M_inv_apply(r)
        return r.clone() if clone else r

    def _solve_single(self):
        A, eps = self.A_apply_function, self.div_eps
        with torch.no_grad():
            x = self.x0.clone()
            r = self.b - A(x)
            z = self._precond(r, True)
            p = z.clone()
            rz = self._inner_product(r, z).real
            bn = torch.linalg.norm(self.b).real
            den = bn if bn > 0 else torch.tensor(1.0, device=self.device, dtype=self.dtype)
            done = 0
            for i in range(self.max_iter):
                done = i + 1
                Ap = A(p)
                alpha = rz / (self._inner_product(p, Ap).real + eps)
                x = x + alpha * p
                r = r - alpha * Ap
                if self.early_stopping and torch.linalg.norm(r).real / (den + eps) < self.tol:
                    break
                z = self._precond(r, False)
                rz_next = self._inner_product(r, z).real
                p = z + (rz_next / (rz + eps)) * p
                rz = rz_next
            self.iters_completed = done
            return x

    def _solve_batched(self):
        A, eps = self.A_apply_function, self.div_eps
        with torch.no_grad():
            x = self.x0.clone()
            r = self.b - A(x)
            z = self._precond(r, True)
            p = z.clone()
            rz = torch.sum(r.conj() * z, dim=1).real
            bn = torch.linalg.norm(self.b, dim=1).real
            den = torch.where(bn > 0, bn, torch.ones_like(bn))
            live = torch.ones(x.shape[0], dtype=torch.bool, device=self.device)
            done = 0
            for i in range(self.max_iter):
                done = i + 1
                idx = torch.where(live)[0]
                if idx.numel() == 0:
                    break
                Ap = A(p[idx])
                alpha = rz[idx] / (torch.sum(p[idx].conj() * Ap, dim=1).real + eps)
                x[idx] += alpha.unsqueeze(1) * p[idx]
                r[idx] -= alpha.unsqueeze(1) * Ap
                z = self._precond(r[idx], False)
                rz_next = torch.sum(r[idx].conj() * z, dim=1).real
                p[idx] = z + (rz_next / (rz[idx] + eps)).unsqueeze(1) * p[idx]
                rz[idx] = rz_next
                if self.early_stopping:
                    rn = torch.linalg.norm(r[idx], dim=1).real
                    stop = (rn / (den[idx] + eps) < self.tol) | (rn < 1e-12)
                    live[idx[stop]] = False
            self.iters_completed = done
            return x

    def _inner_product(self, a, b):
        if a.dim() == 1:
            return torch.dot(torch.conj(a), b)
        if a.dim() == 2:
            return torch.sum(a.conj() * b, dim=1)
        raise ValueError(f"Unsupported tensor dimension: {a.dim()}")
