import sys, os, ast, json
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import test_gpu_e2e as T
from monoflex_amd import synthetic as S
g = np.load(os.path.join(T.ROOT, "tests", "golden", "e2e_full.npz"))
meta = ast.literal_eval(str(g["meta"]))
m = T._hip_model(meta["cls_bias"], "bf16")
imgs = S.synthetic_images(8, 384, 1280, seed=meta["seeds"][0])
det, topk, valid, hm = T._run(m, imgs, [S.synthetic_target(320, 96)] * 8)
mine, ref = topk[0][:, 1].numpy().astype(np.int64), g["img0_topk_index"]
mcls = topk[0][:, 2].numpy()
ref_rows = {int(i): r for i, r in zip(ref[:len(g["img0_result"])], g["img0_result"])}
rows = det[0][valid[0].bool()].numpy()
np.set_printoptions(precision=3, suppress=True, linewidth=200)
for k, (i, r) in enumerate(zip(mine, rows)):
    if int(i) in ref_rows:
        d = np.abs(r - ref_rows[int(i)]) / np.maximum(np.abs(ref_rows[int(i)]), 1.0)
        if d.max() > 0.1:
            print(k, i, mcls[k], "col", d.argmax(), d.max()); print(" mine", r); print(" ref ", ref_rows[int(i)])
print("dup pixel indices in ref:", len(ref) - len(set(ref.tolist())), "in mine:", len(mine) - len(set(mine.tolist())))
