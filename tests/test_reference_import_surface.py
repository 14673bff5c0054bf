"""`runs/monoflex.yaml` and `tools/plain_train_net.py` drive this build unchanged (BASELINE.json north_star): the reference
script's own import statements (tools/plain_train_net.py:9-26, restated here as data) resolve against the repository root,
every imported name exists, and the top-level names are the SAME modules as monoflex_amd.* (no second copies)."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (module, names) of every project import of the reference's entry script
REFERENCE_IMPORTS = [
    ("config", ["cfg"]),
    ("data", ["make_data_loader", "build_test_loader"]),
    ("solver", ["build_optimizer", "build_scheduler"]),
    ("utils.check_point", ["DetectronCheckpointer"]),
    ("engine", ["default_argument_parser", "default_setup", "launch"]),
    ("utils", ["comm"]),
    ("utils.backup_files", ["sync_root"]),
    ("engine.trainer", ["do_train"]),
    ("engine.test_net", ["run_test"]),
    ("model.detector", ["KeypointDetector"]),
    ("model.backbone", ["build_backbone"]),
    ("model.backbone.DCNv2.dcn_v2", ["DCN", "DCNv2", "dcn_v2_conv", "_DCNv2"]),
]


def test_reference_script_imports_resolve_from_the_repo_root():
    code = ["import sys; sys.path.insert(0, %r)" % ROOT]
    for mod, names in REFERENCE_IMPORTS:
        if mod == "utils" and names == ["comm"]:
            code.append("from utils import comm")
        else:
            code.append("from %s import %s" % (mod, ", ".join(names)))
    code += ["import monoflex_amd.model.detector as real, model.detector as alias",
             "assert alias is real and KeypointDetector is real.KeypointDetector",
             "import utils.check_point, monoflex_amd.utils.check_point",
             "assert sys.modules['utils.check_point'] is sys.modules['monoflex_amd.utils.check_point']",
             "cfg.merge_from_file(%r)" % os.path.join(ROOT, "runs", "monoflex.yaml"),
             "cfg.merge_from_list(['SOLVER.IMS_PER_BATCH', '8'])",
             "cfg.DATALOADER.NUM_WORKERS = 2; cfg.TEST.EVAL_DIS_IOUS = False; cfg.START_TIME = 'now'",
             "assert cfg.MODEL.HEAD.NUM_CHANNEL == 256 and cfg.SOLVER.IMS_PER_BATCH == 8",
             "args = default_argument_parser().parse_args(['--config', 'runs/monoflex.yaml', '--batch_size', '8', '--num_gpus', '4'])",
             "assert args.config_file == 'runs/monoflex.yaml' and args.num_gpus == 4 and args.dist_url == 'auto'",
             "assert comm.get_world_size() == 1 and comm.get_rank() == 0 and comm.is_main_process()",
             "print('surface ok')"]
    r = subprocess.run([sys.executable, "-c", "\n".join(code)], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "surface ok" in r.stdout, r.stderr[-2000:]


def test_reference_call_forms_of_solver_and_launch(tmp_path):
    import torch
    sys.path.insert(0, ROOT)
    from monoflex_amd.config import get_cfg
    from monoflex_amd.solver import build_scheduler
    from monoflex_amd.engine.launch import launch
    from monoflex_amd.utils.backup_files import sync_root
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.SOLVER.STEPS = [3, 5]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(2))], lr=cfg.SOLVER.BASE_LR)
    sched, warm = build_scheduler(opt, total_iters_each_epoch=10, optim_cfg=cfg.SOLVER)       # plain_train_net.py:53-56
    lrs = []
    for _ in range(6):
        opt.step(); sched.step(); lrs.append(opt.param_groups[0]["lr"])
    assert warm is None and abs(lrs[1] - cfg.SOLVER.BASE_LR) < 1e-12 and abs(lrs[2] - cfg.SOLVER.BASE_LR * 0.1) < 1e-12
    assert abs(lrs[5] - cfg.SOLVER.BASE_LR * 0.01) < 1e-12
    seen = []
    launch(lambda a, b: seen.append(a + b), num_gpus_per_machine=1, args=(1, 2))               # world 1: called in-process
    assert seen == [3]
    n = sync_root(os.path.join(ROOT, "runs"), str(tmp_path / "backup"))
    assert n >= 1 and os.path.exists(tmp_path / "backup" / "monoflex.yaml")


def _launch_main(tag, out_dir):
    import torch.distributed as dist
    from monoflex_amd.utils import comm
    got = comm.all_gather({"rank": comm.get_rank(), "tag": tag})
    with open(os.path.join(out_dir, "rank%d.txt" % comm.get_rank()), "w") as f:
        f.write("%d %d %d %s" % (comm.get_rank(), comm.get_world_size(), comm.get_local_rank(), sorted(g["rank"] for g in got)))
    assert dist.is_initialized()


def test_launch_spawns_one_process_per_rank_gloo(tmp_path):
    """engine.launch.launch with 2 ranks (gloo on CPU): process group on 127.0.0.1, local group, comm helpers."""
    sys.path.insert(0, ROOT)
    from monoflex_amd.engine.launch import launch
    launch(_launch_main, num_gpus_per_machine=2, dist_url="auto", args=("x", str(tmp_path)), backend="gloo")
    for r in range(2):
        assert open(tmp_path / ("rank%d.txt" % r)).read() == "%d 2 %d [0, 1]" % (r, r)
