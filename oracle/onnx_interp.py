"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

A small ONNX interpreter (torch-CPU fp32) for the operator set of the reference's fixture graphs
/root/reference/tests/stable-diffusion/{vae_decoder,vae_encoder,text_encoder}.onnx (opset 15,
torch 1.13 exports) and of the raw-dialect UNet graph diffusers_amd/modelgen.py writes.  It stands
in for the ``session.run(...)`` ONNX Runtime calls of
/root/reference/src/pipelines/stable_diffusion/impl_txt2img.rs:283 and impl_main.rs:403 — ONNX
Runtime itself is not available in this image.  Operator semantics follow the ONNX opset-15
operator specification.

It is the one way this repo can execute a graph the reference ships.  It pins
oracle/unet_ref.py::vae_decode (tests/test_oracle_vae_fixture.py); its own operator semantics are
pinned on the three real torch-1.13 exports against independent torch restatements
(oracle/fixture_models.py, tests/test_oracle_fixture_graphs.py), and it is then the second route from
weights to UNet output that oracle/unet_ref.py is checked against (tests/test_oracle_unet_graph.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import onnx_io

_TORCH_DT = {1: torch.float32, 6: torch.int32, 7: torch.int64, 9: torch.bool, 10: torch.float16, 11: torch.float64}


def _ints(t) -> list:
    return [int(v) for v in t.reshape(-1).tolist()]


def run(model: onnx_io.Model, feeds: Dict[str, np.ndarray], random_normal=None) -> Dict[str, np.ndarray]:
    """``random_normal(shape) -> np.ndarray``: what RandomNormalLike nodes return (the VAE encoder's posterior sample); zeros if None."""
    env: Dict[str, torch.Tensor] = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in feeds.items()}
    for name, t in model.initializers.items():
        env[name] = torch.from_numpy(t.numpy(model.base_dir))
    env[""] = None
    for n in model.nodes:
        i = [env[x] for x in n.inputs]
        op = n.op_type
        if op == "Identity":
            o = i[0]
        elif op == "Constant":
            o = torch.from_numpy(n.attrs["value"].t.numpy(model.base_dir))
        elif op == "Cast":
            o = i[0].to(_TORCH_DT[n.attr_i("to")])
        elif op == "Conv":
            pads = n.attr_ints("pads", [0, 0, 0, 0])
            assert pads[0] == pads[2] and pads[1] == pads[3] and n.attr_i("group", 1) == 1
            o = F.conv2d(i[0], i[1], i[2] if len(i) > 2 else None, stride=tuple(n.attr_ints("strides", [1, 1])),
                         padding=(pads[0], pads[1]), dilation=tuple(n.attr_ints("dilations", [1, 1])))
        elif op == "Reshape":
            shape = [int(s) for s in i[1].tolist()]
            shape = [i[0].shape[k] if s == 0 else s for k, s in enumerate(shape)]  # allowzero=0
            o = i[0].reshape(shape)
        elif op == "InstanceNormalization":
            o = F.instance_norm(i[0], weight=i[1], bias=i[2], eps=n.attr_f("epsilon", 1e-5))
        elif op == "Shape":
            o = torch.tensor(list(i[0].shape), dtype=torch.int64)
        elif op == "Gather":
            axis = n.attr_i("axis", 0)
            idx = i[1]
            o = torch.index_select(i[0], axis, idx.reshape(-1).long()).reshape(
                list(i[0].shape[:axis]) + list(idx.shape) + list(i[0].shape[axis + 1:]))
        elif op == "Unsqueeze":
            o = i[0]
            for ax in sorted(int(a) for a in i[1].reshape(-1).tolist()):
                o = o.unsqueeze(ax)
        elif op == "Concat":
            o = torch.cat(i, dim=n.attr_i("axis"))
        elif op == "Mul":
            o = i[0] * i[1]
        elif op == "Add":
            o = i[0] + i[1]
        elif op == "Sub":
            o = i[0] - i[1]
        elif op == "Div":
            if i[0].dtype in (torch.int64, torch.int32) and i[1].dtype in (torch.int64, torch.int32):
                o = torch.div(i[0], i[1], rounding_mode="trunc")
            else:
                o = i[0] / i[1]
        elif op == "Sigmoid":
            o = torch.sigmoid(i[0])
        elif op == "Transpose":
            o = i[0].permute(n.attr_ints("perm"))
        elif op == "MatMul":
            o = i[0] @ i[1]
        elif op == "Softmax":
            o = torch.softmax(i[0], dim=n.attr_i("axis", -1))
        elif op == "ConstantOfShape":
            v = n.attrs["value"].t.numpy(model.base_dir) if "value" in n.attrs else np.zeros(1, np.float32)
            o = torch.full([int(s) for s in i[0].tolist()], v.reshape(-1)[0].item(), dtype=torch.from_numpy(v).dtype)
        elif op == "Resize":
            assert n.attr_s("mode") == "nearest" and n.attr_s("coordinate_transformation_mode") == "asymmetric"
            assert n.attr_s("nearest_mode") == "floor"
            if len(i) > 3 and i[3] is not None:          # `sizes`: x_in = floor(x_out * (in / out))  (asymmetric: x_out / scale)
                sizes = _ints(i[3])
                assert sizes[0] == i[0].shape[0] and sizes[1] == i[0].shape[1]
                Hi, Wi = i[0].shape[2], i[0].shape[3]
                ys = torch.floor(torch.arange(sizes[2], dtype=torch.float32) / (torch.tensor(sizes[2], dtype=torch.float32) / Hi)).long().clamp(max=Hi - 1)
                xs = torch.floor(torch.arange(sizes[3], dtype=torch.float32) / (torch.tensor(sizes[3], dtype=torch.float32) / Wi)).long().clamp(max=Wi - 1)
                o = i[0][:, :, ys][:, :, :, xs]
            else:
                scales = i[2].tolist()
                assert scales[0] == 1 and scales[1] == 1
                sh, sw = int(scales[2]), int(scales[3])
                o = i[0].repeat_interleave(sh, dim=2).repeat_interleave(sw, dim=3)
        elif op in ("Sin", "Cos", "Erf", "Exp", "Sqrt", "Tanh"):
            o = getattr(torch, op.lower())(i[0])
        elif op == "Pow":
            o = torch.pow(i[0], i[1])
        elif op == "ReduceMean":
            o = i[0].mean(dim=tuple(n.attr_ints("axes")), keepdim=bool(n.attr_i("keepdims", 1)))
        elif op == "Gemm":
            a = i[0].t() if n.attr_i("transA", 0) else i[0]
            b = i[1].t() if n.attr_i("transB", 0) else i[1]
            o = n.attr_f("alpha", 1.0) * (a @ b)
            if len(i) > 2 and i[2] is not None:
                o = o + n.attr_f("beta", 1.0) * i[2]
        elif op == "Slice":
            starts, ends = _ints(i[1]), _ints(i[2])
            axes = _ints(i[3]) if len(i) > 3 and i[3] is not None else list(range(len(starts)))
            steps = _ints(i[4]) if len(i) > 4 and i[4] is not None else [1] * len(starts)
            o = i[0]
            for st, en, ax, sp in zip(starts, ends, axes, steps):
                dim = o.shape[ax]
                st = st + dim if st < 0 else st
                en = en + dim if en < 0 else en
                if sp > 0:
                    st, en = min(max(st, 0), dim), min(max(en, 0), dim)
                else:          # negative step: start in [0, dim-1], end in [-1, dim-1] (opset-13 Slice)
                    st, en = min(max(st, 0), dim - 1), min(max(en, -1), dim - 1)
                o = torch.index_select(o, ax, torch.arange(st, en, sp, dtype=torch.long))
        elif op == "Pad":
            assert n.attr_s("mode", "constant") == "constant"
            pads = _ints(i[1]); r = i[0].dim()
            val = float(i[2]) if len(i) > 2 and i[2] is not None else 0.0
            tp = []
            for ax in reversed(range(r)):      # torch wants the last dim first
                tp += [pads[ax], pads[ax + r]]
            o = F.pad(i[0], tp, value=val)
        elif op == "Clip":
            lo = i[1] if len(i) > 1 and i[1] is not None else None
            hi = i[2] if len(i) > 2 and i[2] is not None else None
            o = torch.clamp(i[0], min=lo, max=hi)
        elif op == "RandomNormalLike":
            shp = tuple(i[0].shape)
            z = torch.zeros(shp, dtype=torch.float32) if random_normal is None else torch.from_numpy(np.asarray(random_normal(shp), np.float32))
            o = z * n.attr_f("scale", 1.0) + n.attr_f("mean", 0.0)
        elif op == "Trilu":
            k = int(i[1]) if len(i) > 1 and i[1] is not None else 0
            o = torch.triu(i[0], diagonal=k) if n.attr_i("upper", 1) else torch.tril(i[0], diagonal=k)
        elif op == "Range":
            o = torch.arange(i[0].item(), i[1].item(), i[2].item(), dtype=i[0].dtype)
        elif op == "ArgMax":
            o = torch.argmax(i[0], dim=n.attr_i("axis", 0), keepdim=bool(n.attr_i("keepdims", 1)))
        elif op == "Flatten":
            ax = n.attr_i("axis", 1)
            o = i[0].reshape(int(np.prod(i[0].shape[:ax], dtype=np.int64)), -1)
        elif op == "Expand":
            o = i[0].expand(torch.broadcast_shapes(tuple(i[0].shape), tuple(_ints(i[1]))))
        elif op == "Squeeze":
            o = i[0]
            for ax in sorted((int(a) % i[0].dim() for a in _ints(i[1])), reverse=True):
                o = o.squeeze(ax)
        else:
            raise NotImplementedError(f"op {op} ({n.name})")
        env[n.outputs[0]] = o
    return {k: env[k].numpy() for k in model.outputs}
