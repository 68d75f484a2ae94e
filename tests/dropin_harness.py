"""TEST INFRASTRUCTURE: run the reference's launch scripts `main_nce.py` / `main_coclr.py` UNMODIFIED
(imported from /root/reference as they lie there) on top of either this repository's shadow packages
(`model/`, `backbone/` -> coclr_amd) or the reference's own `model/`, `backbone/` (to record the
fixture), through their real `main_worker()`: model construction, `.cuda(gpu)`, the
DistributedDataParallel wrap, one-param-group-per-tensor Adam, `nn.CrossEntropyLoss`, the reference's
FastDataLoader + DistributedSampler, `train_one_epoch`, `calc_topk_accuracy` / `calc_mask_accuracy`,
the `.item()` meters, checkpoint save, `sys.exit(0)`.

What the harness supplies, none of it part of the path under test:
  * modules this image does not have and the scripts import at the top (`torchvision`,
    `tensorboardX`, `lmdb`): minimal stand-ins (`transforms.Compose` is the real three-line class);
  * the data: `get_data` / `get_transform` of the script are replaced by a synthetic dataset (the real
    ones open LMDB files); everything downstream of the dataset is the script's own code;
  * utils/utils.py:67 calls `.view(-1)` on a non-contiguous slice, which PyTorch >= 1.7 rejects (the
    reference pins 1.4): Tensor.view falls back to reshape for such calls;
  * with `cpu=True` (no GPU in the build container): `.cuda()` is the identity, set_device a no-op and
    DistributedDataParallel drops `device_ids` for host modules; the caller passes `--dist-backend gloo`.

Observation points: the logits / targets the script hands to its accuracy helpers (wrapped in the
script's namespace), the losses its AverageMeter receives, and the checkpoint file it writes.
"""
import contextlib
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SHADOWED = ("model", "backbone", "utils", "dataset")


def reference_available():
    return os.path.isfile(os.path.join(REF, "main_nce.py"))


class _Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _Writer:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, *a, **k):
        self.scalars.append(a)

    def add_image(self, *a, **k):
        pass


def _stand_ins():
    mods = {}
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.Compose = _Compose
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    tv.utils = types.ModuleType("torchvision.utils")
    mods.update({"torchvision": tv, "torchvision.transforms": tv.transforms,
                 "torchvision.transforms.functional": tv.transforms.functional,
                 "torchvision.utils": tv.utils})
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = _Writer
    mods["tensorboardX"] = tbx
    mods["lmdb"] = types.ModuleType("lmdb")
    return mods


_DDP_SHIM = [False]


def _host_ddp_shim():
    """DistributedDataParallel(module, device_ids=[gpu]) with a HOST module (no GPU here): drop
    device_ids.  Installed once and left in place (a no-op for device modules): the product's own
    default-flag wrapper (coclr_amd/parallel.py) may be stacked on top of it."""
    if _DDP_SHIM[0]:
        return
    ddp = torch.nn.parallel.DistributedDataParallel
    init = ddp.__init__

    def ddp_init(self, module, *a, **k):
        if not any(p.is_cuda for p in module.parameters()):
            k.pop("device_ids", None)
        return init(self, module, *a, **k)
    ddp_init.__wrapped__ = init
    ddp.__init__ = ddp_init
    _DDP_SHIM[0] = True


@contextlib.contextmanager
def script_environment(use_reference_model, cpu):
    """sys.path / sys.modules arranged so that `utils`, `dataset` come from the reference and
    `model`, `backbone` from the reference (fixture) or from this repository (the product)."""
    saved_path = list(sys.path)
    saved_mods = {k: v for k, v in sys.modules.items()
                  if k.split(".")[0] in _SHADOWED + ("torchvision", "tensorboardX", "lmdb")}
    for k in list(saved_mods):
        del sys.modules[k]
    added = []
    for k, v in _stand_ins().items():
        if k not in sys.modules:
            sys.modules[k] = v
            added.append(k)
    sys.path[:] = ([REF] if use_reference_model else [REPO, REF]) + \
        [p for p in saved_path if os.path.abspath(p or ".") not in (REPO, REF)]
    undo = []

    def patch(obj, name, val):
        undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, val)

    view = torch.Tensor.view

    def lenient_view(self, *a, **k):
        try:
            return view(self, *a, **k)
        except RuntimeError:
            return self.reshape(*a, **k)
    patch(torch.Tensor, "view", lenient_view)
    if cpu:
        patch(torch.Tensor, "cuda", lambda self, *a, **k: self)
        patch(torch.nn.Module, "cuda", lambda self, *a, **k: self)
        patch(torch.cuda, "set_device", lambda *a, **k: None)
        patch(torch.cuda, "device_count", lambda: 1)
        _host_ddp_shim()
    import builtins
    real_print = builtins.print
    try:
        yield
    finally:
        builtins.print = real_print           # main_worker silences print on ranks != 0
        for obj, name, val in reversed(undo):
            setattr(obj, name, val)
        for k in [k for k in sys.modules if k.split(".")[0] in _SHADOWED]:
            del sys.modules[k]
        for k in added:
            sys.modules.pop(k, None)
        sys.modules.update(saved_mods)
        sys.path[:] = saved_path


def load_script(name):
    """Import /root/reference/<name>.py as it lies there (inside script_environment)."""
    path = os.path.join(REF, name + ".py")
    spec = importlib.util.spec_from_file_location("_ref_script_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.__file__ == path
    return mod


class SyntheticClips(torch.utils.data.Dataset):
    """What the LMDB datasets hand to the loop (dataset/lmdb_dataset.py:500-517): ToTensor-like frames
    (C, num_seq*seq_len, H, W) in [0,1) and a label -- or, two-stream, ([rgb, flow], vname, label)."""

    def __init__(self, n, seq_len, img_dim, two_stream, seed, n_sources=7):
        g = torch.Generator().manual_seed(seed)
        self.two_stream = two_stream
        self.frames = [torch.rand(n, 3, 2 * seq_len, img_dim, img_dim, generator=g)
                       for _ in range(2 if two_stream else 1)]
        self.label = torch.randint(0, 5, (n,), generator=g)
        self.vname = torch.randint(0, n_sources, (n,), generator=g)

    def __len__(self):
        return self.label.shape[0]

    def __getitem__(self, i):
        if self.two_stream:
            return [f[i] for f in self.frames], self.vname[i], self.label[i]
        return self.frames[0][i], self.label[i]


def write_pretrained(path, infonce_model, seed):
    """A checkpoint in the format main_nce.py saves (main_nce.py:281-288) whose encoders are WELL
    CONDITIONED: conv weights He-normal from a seeded generator (in state-dict order), BatchNorm at
    identity.  At the scripts' own `normal_(0, 0.01)` initialisation a frozen eval-mode sampler maps
    every clip to (almost) the same feature, so CoCLR's top-k mining (model/pretrain.py:405-410)
    decides between similarities that differ by float round-off; `main_coclr.py --pretrain A B`
    loads these instead, as its README recipe does with InfoNCE checkpoints.  Built from parameter
    names and shapes only, so the reference's model and the product's produce the same file."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in infonce_model.state_dict().items():
        v = v.detach().cpu().clone()
        if v.dim() == 5 and k.endswith("weight"):
            fan_in = v.shape[1] * v.shape[2] * v.shape[3] * v.shape[4]
            v = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        sd[k] = v
    # S3D registers every stage twice (Conv_1a.* and block1.0.*, backbone/s3dg.py:145-150): keys that
    # alias one tensor must carry the same values
    live = infonce_model.state_dict()
    by_ptr = {}
    for k, v in live.items():
        if v.dim() == 5:
            by_ptr.setdefault(v.data_ptr(), []).append(k)
    for keys in by_ptr.values():
        for k in keys[1:]:
            sd[k] = sd[keys[0]]
    torch.save({"epoch": 0, "state_dict": sd, "best_acc": 0, "iteration": 1}, path)
    return sd


PRETRAIN_SEEDS = {"rgb": 501, "flow": 502}


def write_pretrained_pair(directory, use_reference_model, product=None):
    """rgb.pth.tar / flow.pth.tar for `main_coclr.py --pretrain`, from the reference's InfoNCE (fixture
    generation) or the product's (tests): same names, shapes and values either way."""
    # the model is built under its own seed (global RNG state restored afterwards): the parameters the
    # re-draw below leaves alone -- the projection head's biases -- must not depend on what ran before
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(4242)
        if product is None:
            with script_environment(use_reference_model, cpu=True):
                import model.pretrain as product_mod
                model = product_mod.InfoNCE("s3d", 128, 8, 0.999, 0.07)
        else:
            model = product.InfoNCE("s3d", 128, 8, 0.999, 0.07)
    for tag, seed in PRETRAIN_SEEDS.items():
        write_pretrained(os.path.join(directory, tag + ".pth.tar"), model, seed)


def run_script(name, argv, dataset, use_reference_model, cpu, workdir, port=29641, before_train=None):
    """main_worker(gpu=0, ngpus_per_node=1, parse_args()) of the unmodified script.  Returns the
    observation record {"outputs", "targets", "losses", "checkpoint"}."""
    rec = {"outputs": [], "targets": [], "losses": []}
    cwd = os.getcwd()
    saved_env = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "RANK")}
    saved_argv = list(sys.argv)
    import torch.distributed as dist
    with script_environment(use_reference_model, cpu):
        mod = load_script(name)
        if use_reference_model:
            assert sys.modules["model.pretrain"].__file__.startswith(REF)
        else:
            assert sys.modules["model.pretrain"].__file__.startswith(REPO)
        assert sys.modules["utils.utils"].__file__.startswith(REF)
        mod.get_transform = lambda mode, args: None
        mod.get_data = lambda transform, mode, args: dataset
        two_stream = name == "main_coclr"
        acc_name = "calc_mask_accuracy" if two_stream else "calc_topk_accuracy"
        inner_acc = getattr(mod, acc_name)

        def observed_acc(output, target, topk=(1,)):
            rec["outputs"].append(output.detach().cpu().clone())
            rec["targets"].append(target.detach().cpu().clone())
            return inner_acc(output, target, topk)
        setattr(mod, acc_name, observed_acc)
        meter = mod.AverageMeter

        class ObservedMeter(meter):
            def update(self, val, n=1, **kw):
                if self.name == "Loss":
                    rec["losses"].append(float(val))
                return meter.update(self, val, n, **kw)
        mod.AverageMeter = ObservedMeter
        if before_train is not None:
            inner_epoch = mod.train_one_epoch

            def train_one_epoch(data_loader, model, *a, **k):
                before_train(model)
                return inner_epoch(data_loader, model, *a, **k)
            mod.train_one_epoch = train_one_epoch
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0")
        os.makedirs(workdir, exist_ok=True)
        os.chdir(workdir)
        sys.argv = [name + ".py"] + list(argv)
        try:
            args = mod.parse_args()
            # what main() does before it hands over to main_worker (main_nce.py:97-117)
            torch.manual_seed(args.seed)
            np.random.seed(args.seed)
            random.seed(args.seed)
            args.distributed = args.world_size > 1 or args.multiprocessing_distributed
            try:
                mod.main_worker(0, 1, args)
                raise AssertionError("main_worker returned without sys.exit(0)")
            except SystemExit as e:
                assert e.code == 0, "the script exited with %r" % (e.code,)
            ckpt = os.path.join(args.model_path, "epoch%d.pth.tar" % (args.epochs - 1))
            rec["checkpoint"] = torch.load(ckpt, map_location="cpu", weights_only=False)
            rec["args"] = {k: v for k, v in vars(args).items()
                           if isinstance(v, (int, float, str, bool, list, type(None)))}
        finally:
            sys.argv = saved_argv
            os.chdir(cwd)
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            if dist.is_initialized():
                dist.destroy_process_group()
    return rec
