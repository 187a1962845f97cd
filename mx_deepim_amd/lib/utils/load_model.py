"""Checkpoint import — mirror of lib/utils/load_model.py:9-68 (`load_checkpoint`, `convert_context`, `load_param`)
over the NDArray-list reader of `ndarray_file.py` (no MXNet here). Arrays stay on the host as numpy until
`deepIM_flownet.bind` uploads and re-lays them out; `convert=True` uploads them to a device context instead."""
from . import ndarray_file


def load_checkpoint(prefix, epoch):
    """"%s-%04d.params" -> (arg_params, aux_params), split on the `arg:` / `aux:` key prefixes."""
    save_dict = ndarray_file.load("%s-%04d.params" % (prefix, epoch))
    arg_params, aux_params = {}, {}
    for key, value in save_dict.items():
        kind, name = key.split(":", 1)
        if kind == "arg":
            arg_params[name] = value
        if kind == "aux":
            aux_params[name] = value
    return arg_params, aux_params


def save_checkpoint(prefix, epoch, arg_params, aux_params=None):
    """What `mx.model.save_checkpoint` writes for the parameter file (used by the tests as the producer)."""
    blob = {"arg:%s" % k: v for k, v in arg_params.items()}
    blob.update({"aux:%s" % k: v for k, v in (aux_params or {}).items()})
    ndarray_file.save("%s-%04d.params" % (prefix, epoch), blob)


def convert_context(params, ctx):
    return {k: (v.copy() if hasattr(v, "context") and v.context is ctx else ctx.array(getattr(v, "asnumpy", lambda: v)(),
                                                                                       dtype=v.dtype))
            for k, v in params.items()}


def load_param(prefix, epoch, convert=False, ctx=None, process=False):
    arg_params, aux_params = load_checkpoint(prefix, epoch)
    if convert:
        if ctx is None:
            from ...runtime import Context
            ctx = Context.default()
        arg_params = convert_context(arg_params, ctx)
        aux_params = convert_context(aux_params, ctx)
    if process:   # drop the "_test" / "_i2r" name suffixes older checkpoints carry (load_model.py:59-65)
        for suffix in ("_test", "_i2r"):
            for k in [k for k in arg_params if suffix in k]:
                arg_params[k.replace(suffix, "")] = arg_params.pop(k)
    return arg_params, aux_params
