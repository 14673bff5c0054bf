import ast, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_e2e as T
from monoflex_amd import synthetic as S
g = np.load("tests/golden/e2e_full.npz"); meta = ast.literal_eval(str(g["meta"]))
imgs = S.synthetic_images(1, 384, 1280, seed=meta["seeds"][0])
pix = torch.as_tensor(g["img0_pix"])
for mode in ("fp32", "fp16x2"):
    m = T._hip_model(meta["cls_bias"], mode)
    det, topk, valid, hm = T._run(m, imgs, [S.synthetic_target(320, 96)])
    errs = T._stage_errors(g, 0, T._stages(m, imgs))
    lg = hm[0][..., :3].permute(2, 0, 1).reshape(3, -1)[:, pix].numpy()
    d = np.abs(lg - g["img0_cls_logits_at"])
    mine, ref = topk[0][:, 1].numpy().astype(np.int64), g["img0_topk_index"]
    print(mode, "dlogit max %.2e rms %.2e" % (d.max(), np.sqrt((d**2).mean())), "order diff at", np.nonzero(mine != ref)[0].tolist(),
          "score err max %.2e" % np.abs(np.sort(topk[0][:, 0].numpy())[::-1] - g["img0_topk_scores"]).max())
    print("   stages", {k: "%.1e" % v[0] for k, v in errs.items()})
