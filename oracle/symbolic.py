"""Computing graph, operators and Taylor-coefficient propagation, in numpy.

ORACLE -- test infrastructure only (see oracle/__init__.py).

Restates, operator by operator, what the reference's symbolic layer does:

* graph IR and topological order      libsanm/symbolic.h:37-296, symbolic.cpp:22-118
* ``TaylorCoeffProp``                  libsanm/symbolic.cpp:142-304
* the 12 operators reachable from FEA graphs
                                        libsanm/oprs/{misc,elem_arith,analytic_unary,
                                        reduce,linalg}.cpp, libsanm/analytic_unary.cpp
* the ``SymbolVar`` sugar              libsanm/oprs.h:14-103, oprs.cpp:16-102

Every operator implements the reference's five ``OperatorMeta`` hooks
(libsanm/symbolic.h:166-219): ``infer_shape`` + ``eval_bias`` (order 0),
``accum_inp_grad`` (reverse-mode Jacobian), ``compute_order_bias`` (order-k
bias: the part of y_k that does not depend on x_k) and ``compute_coeff``
(y_k once x_k is known).

Simplifications that do not change results: Jacobians are always stored FULL
``(T, out_dim, var_size)`` (the reference also has an ELEMWISE storage class,
libsanm/tensor.h:509-601), zero/one storage tags (tensor.h:152-161) are not
modelled, and only batched graphs are supported (what every FEA graph is).
"""
from __future__ import annotations

import math

import numpy as np

from . import tensor_ops as T_


class SANMNumericalError(RuntimeError):
    """libsanm/utils.h:34-50."""


# --------------------------------------------------------------------------
# graph IR
# --------------------------------------------------------------------------
class VarNode:
    def __init__(self, opr, idx):
        self.owner_opr = opr
        self.idx = idx

    @property
    def graph(self):
        return self.owner_opr.graph


class OperatorNode:
    def __init__(self, graph, meta, param, inputs, nr_output):
        self.graph = graph
        self.meta = meta
        self.param = param
        self.inputs = list(inputs)
        self.outputs = [VarNode(self, i) for i in range(nr_output)]

    def output(self, i):
        return self.outputs[i]

    def input(self, i):
        return self.inputs[i]


class ComputingGraph:
    """libsanm/symbolic.h:283-293."""

    def __init__(self):
        self.oprs = []

    def insert_opr(self, meta, param, inputs):
        opr = OperatorNode(self, meta, param, inputs, meta.nr_output(param))
        self.oprs.append(opr)
        return opr


def topo_sort(outputs):
    """Operators needed by ``outputs`` in dependency order (symbolic.cpp:63-118)."""
    order, seen = [], set()

    def visit(opr):
        if id(opr) in seen:
            return
        seen.add(id(opr))
        for v in opr.inputs:
            visit(v.owner_opr)
        order.append(opr)

    for v in outputs:
        visit(v.owner_opr)
    return order


class VarCtx:
    """Per-run state of a variable: libsanm/symbolic.h:39-82."""

    def __init__(self):
        self.shape = None
        self.coeffs = []
        self.cur_order_bias = None
        self.jacobian = None  # (T, odim, size) or None
        self.user = {}
        self.nr_reader = 0

    def get_bias(self, in_coeff):
        return self.coeffs[-1] if in_coeff else self.cur_order_bias

    def set_bias(self, in_coeff, val):
        if in_coeff:
            self.coeffs[-1] = val
        else:
            self.cur_order_bias = val

    def accum_jac(self, g):
        self.jacobian = g.copy() if self.jacobian is None else self.jacobian + g


class ExeCtx:
    def __init__(self):
        self.var2ctx = {}
        self.order = 0

    def get(self, var) -> VarCtx:
        return self.var2ctx[var]


def _bshape(x, like):
    """Broadcast a batched scalar (T,1) against ``like``."""
    if x.shape == like.shape:
        return x
    return x.reshape((x.shape[0],) + (1,) * (like.ndim - 1))


def _size(shape):
    return int(np.prod(shape[1:]))


# --------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------
class OperatorMeta:
    name = "?"

    def nr_output(self, param):
        return 1

    def infer_shape_eval_bias(self, opr, ctx):
        self.infer_shape(opr, ctx)
        self.eval_bias(opr, ctx)


class PlaceholderOprMeta(OperatorMeta):
    """libsanm/oprs/misc.cpp:13-44."""
    name = "placeholder"

    def infer_shape(self, opr, ctx):
        o = ctx.get(opr.output(0))
        assert len(o.coeffs) == 1
        o.shape = o.coeffs[0].shape

    def eval_bias(self, opr, ctx):
        pass

    def accum_inp_grad(self, opr, ctx):
        pass

    def compute_order_bias(self, opr, ctx):
        o = ctx.get(opr.output(0))
        o.cur_order_bias = np.zeros(o.shape)

    def compute_coeff(self, opr, ctx):
        assert len(ctx.get(opr.output(0)).coeffs) == ctx.order + 1


class ConstantOprMeta(OperatorMeta):
    """libsanm/oprs/misc.cpp:48-100 (the shard slicing is done by the caller)."""
    name = "constant"

    def infer_shape(self, opr, ctx):
        ctx.get(opr.output(0)).shape = opr.param["val"].shape

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(opr.param["val"])

    def accum_inp_grad(self, opr, ctx):
        pass

    def compute_order_bias(self, opr, ctx):
        o = ctx.get(opr.output(0))
        o.cur_order_bias = np.zeros(o.shape)

    def compute_coeff(self, opr, ctx):
        o = ctx.get(opr.output(0))
        o.coeffs.append(np.zeros(o.shape))


def _infer_shape_elemwise(opr, ctx):
    """libsanm/oprs/elem_arith.cpp:13-38: only batched scalars broadcast."""
    oshp = None
    for v in opr.inputs:
        ishp = ctx.get(v).shape
        if oshp is None:
            oshp = ishp
        elif oshp != ishp:
            o_scalar = _size(oshp) == 1
            i_scalar = _size(ishp) == 1
            assert o_scalar or i_scalar, f"invalid shape in elem arith: {oshp} vs {ishp}"
            if o_scalar and not i_scalar:
                oshp = ishp
            assert oshp[0] == ishp[0]
    ctx.get(opr.output(0)).shape = oshp


class LinearCombinationOprMeta(OperatorMeta):
    """libsanm/oprs/elem_arith.cpp:42-124."""
    name = "linear_combination"

    def infer_shape(self, opr, ctx):
        _infer_shape_elemwise(opr, ctx)

    def eval_bias(self, opr, ctx):
        o = ctx.get(opr.output(0))
        res = np.full(o.shape, float(opr.param["bias"]))
        for c, v in zip(opr.param["coeffs"], opr.inputs):
            res = res + c * _bshape(ctx.get(v).coeffs[0], res)
        o.coeffs.append(res)

    def accum_inp_grad(self, opr, ctx):
        o = ctx.get(opr.output(0))
        for c, v in zip(opr.param["coeffs"], opr.inputs):
            i = ctx.get(v)
            if i.shape == o.shape:
                i.accum_jac(o.jacobian * c)
            else:
                i.accum_jac(o.jacobian.sum(axis=2, keepdims=True) * c)

    def _bias(self, opr, ctx, in_coeff):
        o = ctx.get(opr.output(0))
        dst = np.zeros(o.shape)
        for c, v in zip(opr.param["coeffs"], opr.inputs):
            dst = dst + c * _bshape(ctx.get(v).get_bias(in_coeff), dst)
        o.set_bias(in_coeff, dst)

    def compute_order_bias(self, opr, ctx):
        self._bias(opr, ctx, False)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(None)
        self._bias(opr, ctx, True)


class MultiplyOprMeta(OperatorMeta):
    """libsanm/oprs/elem_arith.cpp:128-217 (Cauchy product with broadcast)."""
    name = "multiply"

    def infer_shape(self, opr, ctx):
        _infer_shape_elemwise(opr, ctx)

    @staticmethod
    def _mul(a, b, oshape):
        like = np.empty(oshape)
        return _bshape(a, like) * _bshape(b, like)

    def eval_bias(self, opr, ctx):
        o = ctx.get(opr.output(0))
        o.coeffs.append(self._mul(ctx.get(opr.input(0)).coeffs[0],
                                  ctx.get(opr.input(1)).coeffs[0], o.shape))

    def accum_inp_grad(self, opr, ctx):
        o = ctx.get(opr.output(0))
        ic = [ctx.get(opr.input(0)), ctx.get(opr.input(1))]
        for k in range(2):
            other = ic[1 - k].coeffs[0].reshape(o.shape[0], -1)  # (T, 1|osz)
            gi = o.jacobian * other[:, None, :]
            if ic[k].shape == o.shape:
                ic[k].accum_jac(gi)
            else:
                ic[k].accum_jac(gi.sum(axis=2, keepdims=True))

    def compute_order_bias(self, opr, ctx):
        o = ctx.get(opr.output(0))
        a, b = ctx.get(opr.input(0)), ctx.get(opr.input(1))
        sb = np.zeros(o.shape)
        for i in range(1, ctx.order):
            sb = sb + self._mul(a.coeffs[i], b.coeffs[ctx.order - i], o.shape)
        o.user["self_bias"] = sb
        o.cur_order_bias = (sb + self._mul(a.coeffs[0], b.cur_order_bias, o.shape)
                            + self._mul(a.cur_order_bias, b.coeffs[0], o.shape))

    def compute_coeff(self, opr, ctx):
        o = ctx.get(opr.output(0))
        a, b = ctx.get(opr.input(0)), ctx.get(opr.input(1))
        o.coeffs.append(o.user["self_bias"] + self._mul(a.coeffs[0], b.coeffs[-1], o.shape)
                        + self._mul(a.coeffs[-1], b.coeffs[0], o.shape))


class AnalyticUnaryOprMeta(OperatorMeta):
    """libsanm/oprs/analytic_unary.cpp:113-158 + libsanm/analytic_unary.cpp:13-139.

    param: {"kind": "log"} or {"kind": "pow", "exp": p}.
    """
    name = "analytic_unary"

    def infer_shape(self, opr, ctx):
        ctx.get(opr.output(0)).shape = ctx.get(opr.input(0)).shape

    @staticmethod
    def _eval(p, x):
        if p["kind"] == "log":
            return np.log(x)
        return np.power(x, p["exp"])

    @staticmethod
    def _deriv(p, x):
        if p["kind"] == "log":
            return np.power(x, -1.0)
        return np.power(x, p["exp"] - 1) * p["exp"]

    def eval_bias(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        o.coeffs.append(self._eval(opr.param, i.coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        k = self._deriv(opr.param, i.coeffs[0])
        o.user["k"] = k
        i.accum_jac(o.jacobian * k.reshape(k.shape[0], 1, -1))

    @staticmethod
    def _pow_int(x, exp, k):
        """Repeated-squaring convolution path; analytic_unary.cpp:46-92."""
        def conv(a, b):
            dst = [np.zeros_like(x[0]) for _ in range(k + 1)]
            for i in range(len(a)):
                for j in range(len(b)):
                    if i + j <= k:
                        dst[i + j] = dst[i + j] + a[i] * b[j]
            return dst

        def conv_k(a, b):
            acc = np.zeros_like(x[0])
            for i in range(max(0, k + 1 - len(b)), min(len(a), k + 1)):
                acc = acc + a[i] * b[k - i]
            return acc

        xi, prod = list(x), None
        exp = int(exp)
        while exp > 1:
            if exp % 2:
                prod = list(xi) if prod is None else conv(prod, xi)
            if exp == 2 and prod is None:
                return conv_k(xi, xi)
            exp //= 2
            xi = conv(xi, xi)
        assert prod is not None
        return conv_k(prod, xi)

    def _prop(self, opr, f, x, user):
        """UnaryAnalyticTrait::prop_taylor_coeff; analytic_unary.cpp:148-159."""
        p = opr.param
        k = len(f)
        if k == 1:
            return np.zeros_like(f[0])
        if p["kind"] == "log":
            dst = np.zeros_like(f[0])
            for i in range(1, k):
                dst = dst + x[k - i] * f[i] * (-float(i) / float(k))
            return dst / x[0]
        e = p["exp"]
        if "has_zero" not in user:
            hz = bool((np.abs(x[0]) < 1e-3).any())
            if hz and (e <= 0.5 or math.floor(e) != e):
                raise SANMNumericalError(f"0^p when p is not integer: {e}")
            user["has_zero"] = hz
        if user["has_zero"]:
            return self._pow_int(x, e, k)
        dst = np.zeros_like(f[0])
        for i in range(1, k):
            dst = dst + f[k - i] * x[i] * (float(i) / float(k) * (e + 1) - 1)
        return dst / x[0]

    def compute_order_bias(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        sb = self._prop(opr, o.coeffs, i.coeffs, o.user)
        o.user["self_bias"] = sb
        o.cur_order_bias = o.user["k"] * i.cur_order_bias + sb

    def compute_coeff(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        o.coeffs.append(i.coeffs[-1] * o.user["k"] + o.user["self_bias"])


class ReduceOprMeta(OperatorMeta):
    """Sum over axis; libsanm/oprs/reduce.cpp:11-102.  axis=-1: all non-batch."""
    name = "reduce"

    def _red(self, opr, x):
        ax = opr.param["axis"]
        if ax == -1:
            return x.reshape(x.shape[0], -1).sum(axis=1)[:, None]
        return x.sum(axis=ax, keepdims=opr.param["keepdim"])

    def infer_shape(self, opr, ctx):
        ishp = ctx.get(opr.input(0)).shape
        ctx.get(opr.output(0)).shape = self._red(opr, np.empty(ishp)).shape

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(self._red(opr, ctx.get(opr.input(0)).coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        T, odim, osz = o.jacobian.shape
        ax = opr.param["axis"]
        if ax == -1:
            g = np.broadcast_to(o.jacobian, (T, odim, _size(i.shape)))
        else:
            oshape_keep = list(i.shape)
            oshape_keep[ax] = 1
            g = np.broadcast_to(o.jacobian.reshape([T, odim] + oshape_keep[1:]),
                                [T, odim] + list(i.shape[1:])).reshape(T, odim, -1)
        i.accum_jac(np.ascontiguousarray(g))

    def compute_order_bias(self, opr, ctx):
        ctx.get(opr.output(0)).cur_order_bias = self._red(opr, ctx.get(opr.input(0)).cur_order_bias)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(self._red(opr, ctx.get(opr.input(0)).coeffs[-1]))


def _mm_convolution(x, y, trans_x=False, trans_y=False, order=None):
    """sum_{i} X_i Y_{order-i} over the *known* terms; oprs/linalg.cpp:14-40."""
    if order is None:
        order = len(x)
        assert order == len(y)
    begin = order - len(y) + 1 if order >= len(y) else 0
    end = min(len(x), order + 1)
    dst = None
    for i in range(begin, end):
        t = T_.batched_mm(x[i], y[order - i], trans_x, trans_y)
        dst = t if dst is None else dst + t
    if dst is None:
        dst = np.zeros((x[0].shape[0], x[0].shape[1], y[0].shape[2]))
    return dst


class BatchMatInvMulOprMeta(OperatorMeta):
    """Y X = A (is_left) or X Y = A; libsanm/oprs/linalg.cpp:67-217."""
    name = "batch_mat_inv_mul"

    def infer_shape(self, opr, ctx):
        ctx.get(opr.output(0)).shape = ctx.get(opr.input(0)).shape

    def eval_bias(self, opr, ctx):
        x, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        xinv = T_.batched_matinv(x.coeffs[0])
        o.user["xinv"] = xinv
        if opr.param["use_identity"]:
            o.coeffs.append(xinv)
            return
        a = ctx.get(opr.input(1)).coeffs[0]
        o.coeffs.append(T_.batched_mm(a, xinv) if opr.param["is_left"] else T_.batched_mm(xinv, a))

    def accum_inp_grad(self, opr, ctx):
        p = opr.param
        x, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        xinv = o.user["xinv"]
        if p["is_left"]:
            m0, m1 = -o.coeffs[0], xinv
        else:
            m0, m1 = xinv, -o.coeffs[0]
        T, odim, _ = o.jacobian.shape
        n = m0.shape[1]
        gy = o.jacobian.reshape(T, odim, n, n)
        # gx[b,r,(i,j)] = gy[b,r,(p,q)] m0[b,p,i] m1[b,j,q]
        gx = np.einsum("brpq,bpi,bjq->brij", gy, m0, m1).reshape(T, odim, n * n)
        x.accum_jac(gx)
        if not p["use_identity"]:
            if p["is_left"]:
                ga = np.einsum("briq,bjq->brij", gy, xinv)
            else:
                ga = np.einsum("brpj,bpi->brij", gy, xinv)
            ctx.get(opr.input(1)).accum_jac(ga.reshape(T, odim, n * n))

    def _bias(self, opr, ctx, in_coeff):
        p = opr.param
        x, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        sb = o.user["self_bias"]
        tmp0 = sb if p["use_identity"] else ctx.get(opr.input(1)).get_bias(in_coeff) + sb
        if p["is_left"]:
            tmp1 = T_.batched_mm(o.coeffs[0], x.get_bias(in_coeff))
        else:
            tmp1 = T_.batched_mm(x.get_bias(in_coeff), o.coeffs[0])
        tmp1 = tmp0 - tmp1
        xinv = o.user["xinv"]
        o.set_bias(in_coeff, T_.batched_mm(tmp1, xinv) if p["is_left"] else T_.batched_mm(xinv, tmp1))

    def compute_order_bias(self, opr, ctx):
        x, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        if opr.param["is_left"]:
            sb = _mm_convolution(o.coeffs, x.coeffs)
        else:
            sb = _mm_convolution(x.coeffs, o.coeffs)
        # the convolution above includes the i=0 / i=order-? terms only for
        # indices in range; with len == order both ends are excluded
        o.user["self_bias"] = -sb
        self._bias(opr, ctx, False)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(None)
        self._bias(opr, ctx, True)


class BatchDeterminantOprMeta(OperatorMeta):
    """libsanm/oprs/linalg.cpp:221-282."""
    name = "batch_determinant"

    def infer_shape(self, opr, ctx):
        ctx.get(opr.output(0)).shape = (ctx.get(opr.input(0)).shape[0], 1)

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(T_.batched_determinant(ctx.get(opr.input(0)).coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        cof = T_.batched_cofactor(i.coeffs[0])
        o.user["cof"] = cof
        T = cof.shape[0]
        i.accum_jac(np.matmul(o.jacobian, cof.reshape(T, 1, -1)))

    def compute_order_bias(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        sb = T_.compute_polymat_det_coeff(i.coeffs, ctx.order)
        o.user["self_bias"] = sb
        T = sb.shape[0]
        o.cur_order_bias = (i.cur_order_bias.reshape(T, -1) * o.user["cof"].reshape(T, -1)
                            ).sum(axis=1)[:, None] + sb

    def compute_coeff(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        T = i.coeffs[-1].shape[0]
        o.coeffs.append((i.coeffs[-1].reshape(T, -1) * o.user["cof"].reshape(T, -1)
                         ).sum(axis=1)[:, None] + o.user["self_bias"])


class BatchMatTransposeOprMeta(OperatorMeta):
    """libsanm/oprs/linalg.cpp:286-335."""
    name = "batch_mat_transpose"

    def infer_shape(self, opr, ctx):
        s = ctx.get(opr.input(0)).shape
        ctx.get(opr.output(0)).shape = (s[0], s[2], s[1])

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(T_.batched_transpose(ctx.get(opr.input(0)).coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        T, odim, _ = o.jacobian.shape
        d0, d1 = i.shape[1], i.shape[2]
        g = o.jacobian.reshape(T, odim, d1, d0).swapaxes(2, 3).reshape(T, odim, d0 * d1)
        i.accum_jac(np.ascontiguousarray(g))

    def compute_order_bias(self, opr, ctx):
        ctx.get(opr.output(0)).cur_order_bias = T_.batched_transpose(ctx.get(opr.input(0)).cur_order_bias)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(T_.batched_transpose(ctx.get(opr.input(0)).coeffs[-1]))


class BatchMatMulOprMeta(OperatorMeta):
    """libsanm/oprs/linalg.cpp:339-418."""
    name = "batch_mat_mul"

    def infer_shape(self, opr, ctx):
        sl, sr = ctx.get(opr.input(0)).shape, ctx.get(opr.input(1)).shape
        assert sl[0] == sr[0] and sl[2] == sr[1]
        ctx.get(opr.output(0)).shape = (sl[0], sl[1], sr[2])

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(
            T_.batched_mm(ctx.get(opr.input(0)).coeffs[0], ctx.get(opr.input(1)).coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        i0, i1, o = ctx.get(opr.input(0)), ctx.get(opr.input(1)), ctx.get(opr.output(0))
        T, odim, _ = o.jacobian.shape
        m, k, n = i0.shape[1], i0.shape[2], i1.shape[2]
        g = o.jacobian.reshape(T, odim, m, n)
        i0.accum_jac(np.einsum("brmn,bkn->brmk", g, i1.coeffs[0]).reshape(T, odim, m * k))
        i1.accum_jac(np.einsum("brmn,bmk->brkn", g, i0.coeffs[0]).reshape(T, odim, k * n))

    def _bias(self, opr, ctx, in_coeff):
        i0, i1, o = ctx.get(opr.input(0)), ctx.get(opr.input(1)), ctx.get(opr.output(0))
        dst = (T_.batched_mm(i0.get_bias(in_coeff), i1.coeffs[0])
               + T_.batched_mm(i0.coeffs[0], i1.get_bias(in_coeff)) + o.user["self_bias"])
        o.set_bias(in_coeff, dst)

    def compute_order_bias(self, opr, ctx):
        i0, i1, o = ctx.get(opr.input(0)), ctx.get(opr.input(1)), ctx.get(opr.output(0))
        o.user["self_bias"] = _mm_convolution(i0.coeffs, i1.coeffs)
        self._bias(opr, ctx, False)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(None)
        self._bias(opr, ctx, True)


class BatchMulEyeOprMeta(OperatorMeta):
    """s * I_dim; libsanm/oprs/linalg.cpp:422-479."""
    name = "batch_mul_eye"

    def _eye(self, opr, s):
        d = opr.param["dim"]
        return s.reshape(-1, 1, 1) * np.eye(d)[None]

    def infer_shape(self, opr, ctx):
        d = opr.param["dim"]
        ctx.get(opr.output(0)).shape = (ctx.get(opr.input(0)).shape[0], d, d)

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(self._eye(opr, ctx.get(opr.input(0)).coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        T, odim, _ = o.jacobian.shape
        d = opr.param["dim"]
        g = np.einsum("brii->br", o.jacobian.reshape(T, odim, d, d))[:, :, None]
        i.accum_jac(g)

    def compute_order_bias(self, opr, ctx):
        ctx.get(opr.output(0)).cur_order_bias = self._eye(opr, ctx.get(opr.input(0)).cur_order_bias)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(self._eye(opr, ctx.get(opr.input(0)).coeffs[-1]))


class BatchSVDWOprMeta(OperatorMeta):
    """SVD-W with outputs (U, S, W); libsanm/oprs/linalg.cpp:483-615.

    ``pw_mode`` (only W is read: the ARAP graph) keeps the polar factor P_i and
    uses ``svd_w_taylor_fwd_p``; otherwise the full U/S/W recurrences.
    """
    name = "batch_svd_w"

    def nr_output(self, param):
        return 3

    def infer_shape(self, opr, ctx):
        s = ctx.get(opr.input(0)).shape
        ctx.get(opr.output(0)).shape = s
        ctx.get(opr.output(1)).shape = (s[0], s[1])
        ctx.get(opr.output(2)).shape = s

    def eval_bias(self, opr, ctx):
        u, s, w = T_.batched_svd_w(ctx.get(opr.input(0)).coeffs[0], opr.param["require_rotation"])
        for k, v in enumerate((u, s, w)):
            ctx.get(opr.output(k)).coeffs.append(v)

    def accum_inp_grad(self, opr, ctx):
        i = ctx.get(opr.input(0))
        uc, sc, wc = (ctx.get(opr.output(k)) for k in range(3))
        du, ds, dw = T_.svd_w_jacobians(uc.coeffs[0], sc.coeffs[0], wc.coeffs[0],
                                        uc.jacobian is not None, sc.jacobian is not None,
                                        wc.jacobian is not None)
        for c, d in ((uc, du), (sc, ds), (wc, dw)):
            if c.jacobian is not None:
                i.accum_jac(np.matmul(c.jacobian, d))
        wc.user["svd"] = {"P": []}

    def compute_order_bias(self, opr, ctx):
        i = ctx.get(opr.input(0))
        uc, sc, wc = (ctx.get(opr.output(k)) for k in range(3))
        ud = wc.user["svd"]
        z = np.zeros_like(i.coeffs[0])
        if ctx.order == 1:
            for k in ("Bu", "Bw", "Mbiask", "Bm", "Bp", "Bpw"):
                ud[k] = z
            assert not ud["P"]
            ud["pw_mode"] = (uc.nr_reader == 0 and sc.nr_reader == 0)
            if ud["pw_mode"]:
                ud["P"].append(None)  # P0 is not used
                wc.cur_order_bias = z.copy()
            else:
                uc.cur_order_bias = z.copy()
                sc.cur_order_bias = np.zeros_like(sc.coeffs[0])
                wc.cur_order_bias = z.copy()
            return
        k = ctx.order
        if ud["pw_mode"]:
            assert len(ud["P"]) == k
            ud["Bm"] = _mm_convolution(i.coeffs, i.coeffs, False, True)
            ud["Bp"] = _mm_convolution(ud["P"], ud["P"], True, False)
            ud["Bpw"] = _mm_convolution(ud["P"], wc.coeffs)
        else:
            ud["Bu"] = _mm_convolution(uc.coeffs, uc.coeffs, True)
            ud["Bw"] = _mm_convolution(wc.coeffs, wc.coeffs, True)
            # U S, then U S U', then U S U' W, keeping only already-known terms
            tmp0 = self._conv_arr(k, uc.coeffs, sc.coeffs, y_as_diag=True)
            tmp1 = self._conv_arr(k, tmp0, uc.coeffs, trans_y=True)
            ud["Mbiask"] = _mm_convolution(tmp1, wc.coeffs, False, False, k)
        self._bias(opr, ctx, False)

    @staticmethod
    def _conv_arr(order, x, y, trans_y=False, y_as_diag=False):
        """oprs/linalg.cpp:42-62."""
        dst = []
        for i in range(order + 1):
            acc = None
            begin = i - len(y) + 1 if i >= len(y) else 0
            for j in range(begin, min(len(x), i + 1)):
                if y_as_diag:
                    t = x[j] * y[i - j][:, None, :]
                else:
                    t = T_.batched_mm(x[j], y[i - j], False, trans_y)
                acc = t if acc is None else acc + t
            assert acc is not None
            dst.append(acc)
        return dst

    def _bias(self, opr, ctx, in_coeff):
        i = ctx.get(opr.input(0))
        uc, sc, wc = (ctx.get(opr.output(k)) for k in range(3))
        ud = wc.user["svd"]
        if ud["pw_mode"]:
            pk, wk = T_.svd_w_taylor_fwd_p(i.get_bias(in_coeff), uc.coeffs[0], sc.coeffs[0],
                                           wc.coeffs[0], ud["Bm"], ud["Bp"], ud["Bpw"])
            wc.set_bias(in_coeff, wk)
            if in_coeff:
                ud["P"].append(pk)
        else:
            uk, sk, wk = T_.svd_w_taylor_fwd(i.get_bias(in_coeff), ud["Mbiask"], uc.coeffs[0],
                                             sc.coeffs[0], wc.coeffs[0], ud["Bu"], ud["Bw"])
            uc.set_bias(in_coeff, uk)
            sc.set_bias(in_coeff, sk)
            wc.set_bias(in_coeff, wk)

    def compute_coeff(self, opr, ctx):
        uc, sc, wc = (ctx.get(opr.output(k)) for k in range(3))
        if not wc.user["svd"]["pw_mode"]:
            uc.coeffs.append(None)
            sc.coeffs.append(None)
        wc.coeffs.append(None)
        self._bias(opr, ctx, True)


_PLACEHOLDER = PlaceholderOprMeta()
_CONSTANT = ConstantOprMeta()
_LINCOMB = LinearCombinationOprMeta()
_MULTIPLY = MultiplyOprMeta()
_UNARY = AnalyticUnaryOprMeta()
_REDUCE = ReduceOprMeta()
_MATINVMUL = BatchMatInvMulOprMeta()
_DET = BatchDeterminantOprMeta()
_TRANSPOSE = BatchMatTransposeOprMeta()
_MATMUL = BatchMatMulOprMeta()
_MULEYE = BatchMulEyeOprMeta()
_SVDW = BatchSVDWOprMeta()


# --------------------------------------------------------------------------
# SymbolVar sugar: libsanm/oprs.h:14-103, oprs.cpp:16-102
# --------------------------------------------------------------------------
class SliceOprMeta(OperatorMeta):
    """x[:, begin:end] of a batch-1 (1, n) tensor; libsanm/oprs/misc.cpp:104-231 (axis 1, stride 1, one batch:
    what the reference implements)."""
    name = "slice"

    @staticmethod
    def abs_interval(param, size):
        """misc.cpp:104-133."""
        stride = param["stride"]
        assert stride != 0 and size > 0
        begin, end = param["begin"], param["end"]
        if begin is None:
            begin = 0 if stride > 0 else size - 1
        elif begin < 0:
            begin += size
        if end is None:
            end = size if stride > 0 else -1
        elif end < 0:
            end += size
        if stride < 0:
            assert begin > end and end >= -1 and begin < size
        else:
            assert begin < end and begin >= 0 and end <= size
        return begin, end

    def _cut(self, opr, x):
        p = opr.param
        assert p["axis"] == 1 and p["stride"] == 1 and x.shape[0] == 1, "unimplemented (misc.cpp:149-151)"
        b, e = self.abs_interval(p, x.shape[1])
        return x[:, b:e].copy()

    def infer_shape(self, opr, ctx):
        ishp = ctx.get(opr.input(0)).shape
        p = opr.param
        assert 0 <= p["axis"] < len(ishp)
        b, e = self.abs_interval(p, ishp[p["axis"]])
        oshp = list(ishp)
        oshp[p["axis"]] = (abs(e - b) - 1) // abs(p["stride"]) + 1
        ctx.get(opr.output(0)).shape = tuple(oshp)

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(self._cut(opr, ctx.get(opr.input(0)).coeffs[0]))

    def accum_inp_grad(self, opr, ctx):
        """misc.cpp:166-197"""
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        b, e = self.abs_interval(opr.param, i.shape[1])
        T, odim, _ = o.jacobian.shape
        g = np.zeros((T, odim, i.shape[1]))
        g[:, :, b:e] = o.jacobian
        i.accum_jac(g)

    def compute_order_bias(self, opr, ctx):
        i, o = ctx.get(opr.input(0)), ctx.get(opr.output(0))
        o.cur_order_bias = np.zeros(o.shape) if ctx.order == 1 else self._cut(opr, i.cur_order_bias)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(self._cut(opr, ctx.get(opr.input(0)).coeffs[-1]))


class ConcatOprMeta(OperatorMeta):
    """concatenation along axis 1 of batch-1 tensors; libsanm/oprs/misc.cpp:233-331."""
    name = "concat"

    def infer_shape(self, opr, ctx):
        axis = opr.param["axis"]
        oshp = None
        for v in opr.inputs:
            ishp = list(ctx.get(v).shape)
            if oshp is None:
                oshp = ishp
            else:
                assert len(ishp) == len(oshp)
                oshp[axis] += ishp[axis]
                ishp[axis] = oshp[axis]
                assert ishp == oshp, f"concat shape mismatch {ishp} vs {oshp}"
        ctx.get(opr.output(0)).shape = tuple(oshp)

    def _cat(self, opr, ctx, in_coeff):
        o = ctx.get(opr.output(0))
        assert opr.param["axis"] == 1 and o.shape[0] == 1, "unimplemented (misc.cpp:303)"
        o.set_bias(in_coeff, np.concatenate([ctx.get(v).get_bias(in_coeff).reshape(1, -1) for v in opr.inputs], axis=1))

    def eval_bias(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(None)
        self._cat(opr, ctx, True)

    def accum_inp_grad(self, opr, ctx):
        o = ctx.get(opr.output(0))
        off = 0
        for v in opr.inputs:
            i = ctx.get(v)
            n = i.shape[1]
            i.accum_jac(o.jacobian[:, :, off:off + n])
            off += n
        assert off == o.shape[1]

    def compute_order_bias(self, opr, ctx):
        o = ctx.get(opr.output(0))
        if ctx.order == 1:
            o.cur_order_bias = np.zeros(o.shape)
        else:
            self._cat(opr, ctx, False)

    def compute_coeff(self, opr, ctx):
        ctx.get(opr.output(0)).coeffs.append(None)
        self._cat(opr, ctx, True)


_SLICE = SliceOprMeta()
_CONCAT = ConcatOprMeta()


class SymbolVar:
    def __init__(self, var):
        self.var = var.var if isinstance(var, SymbolVar) else var

    def node(self):
        return self.var

    @property
    def _g(self):
        return self.var.graph

    def __add__(self, rhs):
        if isinstance(rhs, SymbolVar):
            return linear_combine([(1.0, self), (1.0, rhs)])
        return linear_combine([(1.0, self)], float(rhs))

    def __sub__(self, rhs):
        if isinstance(rhs, SymbolVar):
            return linear_combine([(1.0, self), (-1.0, rhs)])
        return self + (-float(rhs))

    def __rsub__(self, lhs):
        return linear_combine([(-1.0, self)], float(lhs))

    def __mul__(self, rhs):
        if isinstance(rhs, SymbolVar):
            return SymbolVar(self._g.insert_opr(_MULTIPLY, None, [self.var, rhs.var]).output(0))
        return linear_combine([(float(rhs), self)], 0.0)

    def reduce_sum(self, axis, keepdim=True):
        assert axis != 0, "can not reduce on batch dim"
        return SymbolVar(self._g.insert_opr(_REDUCE, {"axis": axis, "keepdim": keepdim},
                                            [self.var]).output(0))

    def batched_transpose(self):
        return SymbolVar(self._g.insert_opr(_TRANSPOSE, None, [self.var]).output(0))

    def batched_matinv(self):
        return batched_mat_inv_mul(self, None, True)

    def batched_matmul(self, rhs):
        return SymbolVar(self._g.insert_opr(_MATMUL, None, [self.var, rhs.var]).output(0))

    def batched_det(self):
        return SymbolVar(self._g.insert_opr(_DET, None, [self.var]).output(0))

    def batched_mul_eye(self, dim):
        return SymbolVar(self._g.insert_opr(_MULEYE, {"dim": int(dim)}, [self.var]).output(0))

    def pow(self, exp):
        if exp == 1.0:
            return self
        assert abs(exp) > 1e-9, "zero power not handled"
        return SymbolVar(self._g.insert_opr(_UNARY, {"kind": "pow", "exp": float(exp)},
                                            [self.var]).output(0))

    def log(self):
        return SymbolVar(self._g.insert_opr(_UNARY, {"kind": "log"}, [self.var]).output(0))

    def slice(self, axis, begin=None, end=None, stride=1):
        """oprs.h:60 / misc.cpp:222-229"""
        assert axis >= 0 and stride != 0
        return SymbolVar(self._g.insert_opr(_SLICE, {"axis": int(axis), "begin": begin, "end": end, "stride": int(stride)},
                                            [self.var]).output(0))

    def batched_svd_w(self, require_rotation=False):
        opr = self._g.insert_opr(_SVDW, {"require_rotation": bool(require_rotation)}, [self.var])
        return [SymbolVar(opr.output(i)) for i in range(3)]


def batched_mat_inv_mul(x, a, is_left):
    inp = [x.var] + ([a.var] if a is not None else [])
    p = {"is_left": bool(is_left), "use_identity": a is None}
    return SymbolVar(x.var.graph.insert_opr(_MATINVMUL, p, inp).output(0))


def linear_combine(vars_, bias=0.0):
    coeffs = [float(c) for c, _ in vars_]
    inputs = [v.var for _, v in vars_]
    assert inputs
    g = inputs[0].graph
    return SymbolVar(g.insert_opr(_LINCOMB, {"coeffs": coeffs, "bias": float(bias)}, inputs).output(0))


def concat(vars_, axis):
    """oprs.h / misc.cpp:321-331"""
    inputs = [v.var for v in vars_]
    assert inputs and axis >= 0
    return SymbolVar(inputs[0].graph.insert_opr(_CONCAT, {"axis": int(axis)}, inputs).output(0))


def placeholder(cg):
    return SymbolVar(cg.insert_opr(_PLACEHOLDER, None, []).output(0))


def constant(cg, val):
    val = np.ascontiguousarray(val, dtype=np.float64)
    return SymbolVar(cg.insert_opr(_CONSTANT, {"val": val}, []).output(0))


# --------------------------------------------------------------------------
# TaylorCoeffProp: libsanm/symbolic.cpp:142-304
# --------------------------------------------------------------------------
class TaylorCoeffProp:
    def __init__(self, output):
        output = output.var if isinstance(output, SymbolVar) else output
        self.topo = topo_sort([output])
        self.ctx = ExeCtx()
        self.output_var = output
        self.input_vars = []
        for opr in self.topo:
            for v in opr.outputs:
                self.ctx.var2ctx[v] = VarCtx()
        for opr in self.topo:
            for v in opr.inputs:
                self.ctx.get(v).nr_reader += 1
            if opr.meta is _PLACEHOLDER:
                self.input_vars.append(opr.output(0))
        assert self.input_vars, "no input var found"
        self.xi_known = False
        self.jacobian_done = False

    def push_xi(self, inp_vals):
        """symbolic.cpp:162-204."""
        assert not self.xi_known
        if isinstance(inp_vals, np.ndarray):
            inp_vals = [inp_vals]
        assert len(inp_vals) == len(self.input_vars)
        for v, val in zip(self.input_vars, inp_vals):
            self.ctx.get(v).coeffs.append(np.asarray(val, dtype=np.float64))
        for opr in self.topo:
            if self.ctx.order == 0:
                opr.meta.infer_shape_eval_bias(opr, self.ctx)
            else:
                opr.meta.compute_coeff(opr, self.ctx)
        self.xi_known = True
        return self.ctx.get(self.output_var).coeffs[-1]

    def ensure_jacobian(self):
        """Reverse sweep, seeded with the identity; symbolic.cpp:206-247."""
        if self.jacobian_done:
            return
        assert self.ctx.order == 0
        o = self.ctx.get(self.output_var)
        T, sz = o.shape[0], _size(o.shape)
        o.jacobian = np.broadcast_to(np.eye(sz), (T, sz, sz)).copy()
        for opr in reversed(self.topo):
            opr.meta.accum_inp_grad(opr, self.ctx)
        for opr in self.topo:
            if opr.meta is not _PLACEHOLDER:
                for v in opr.outputs:
                    self.ctx.get(v).jacobian = None
        self.jacobian_done = True

    def compute_next_order_bias(self):
        """symbolic.cpp:249-289."""
        self.ensure_jacobian()
        assert self.xi_known
        self.ctx.order += 1
        self.xi_known = False
        for c in self.ctx.var2ctx.values():
            c.cur_order_bias = None
        for opr in self.topo:
            opr.meta.compute_order_bias(opr, self.ctx)
            if self.ctx.order == 1:
                for v in opr.outputs:
                    c = self.ctx.get(v)
                    if c.nr_reader and c.cur_order_bias is not None:
                        assert not np.any(c.cur_order_bias), \
                            f"opr {opr.meta.name}: bias is non-zero for first order"
        return self.ctx.get(self.output_var).cur_order_bias

    def get_jacobian(self, x=None):
        self.ensure_jacobian()
        x = self.input_vars[0] if x is None else (x.var if isinstance(x, SymbolVar) else x)
        return self.ctx.get(x).jacobian

    def coeffs_of(self, var):
        var = var.var if isinstance(var, SymbolVar) else var
        return self.ctx.get(var).coeffs


def eval_unary_func(y, x):
    """libsanm/symbolic.cpp (eval_unary_func): y(x) for a single-input graph."""
    return TaylorCoeffProp(y).push_xi([x])
