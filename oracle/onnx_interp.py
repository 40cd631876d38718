"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

A small ONNX interpreter (torch-CPU fp32) for exactly the operator set of the reference's
fixture graph /root/reference/tests/stable-diffusion/vae_decoder.onnx (opset 15, torch 1.13
export).  It stands in for the ``vae_decoder.run(...)`` ONNX Runtime call of
/root/reference/src/pipelines/stable_diffusion/impl_main.rs:403 — ONNX Runtime itself is not
available in this image.  Operator semantics follow the ONNX opset-15 operator specification.

It is the one way this repo can execute a graph the reference ships, and pins
oracle/unet_ref.py::vae_decode (tests/test_oracle_vae_fixture.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import onnx_io

_TORCH_DT = {1: torch.float32, 6: torch.int32, 7: torch.int64, 9: torch.bool, 10: torch.float16, 11: torch.float64}


def run(model: onnx_io.Model, feeds: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
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
            scales = i[2].tolist()
            assert scales[0] == 1 and scales[1] == 1
            sh, sw = int(scales[2]), int(scales[3])
            o = i[0].repeat_interleave(sh, dim=2).repeat_interleave(sw, dim=3)
        else:
            raise NotImplementedError(f"op {op} ({n.name})")
        env[n.outputs[0]] = o
    return {k: env[k].numpy() for k in model.outputs}
