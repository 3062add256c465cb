"""What the reference's CVC-YOLOv3/train.py does with the model, statement for statement, on one fixture batch -- run UNCHANGED under
`python -m torch.distributed.run --nproc-per-node N` with the drop-in directory in front on PYTHONPATH (tests/test_gpu_dp.py):

    train.py:19        from models import Darknet
    train.py:100       model = Darknet(config_path=..., xy_loss=..., ...)
    train.py:180-187   optimizer = torch.optim.Adam / SGD(filter(lambda p: p.requires_grad, model.parameters()), ...)
    train.py:191       model.load_weights(weights_path, model.get_start_weight_dim())
    train.py:193-195   if torch.cuda.device_count() > 1: model = nn.DataParallel(model)
    train.py:196       model = model.to(device, non_blocking=True)
    train.py:60-72     imgs / targets .to(device) ; optimizer.zero_grad() ; losses = model(imgs, targets) ; losses[0].sum().backward() ; optimizer.step()
    train.py:74-88     loss.sum().to('cpu').item() ; loss.item()

Nothing here knows about ranks, shards or all-reduce: that is the point.  usage: train_loop_stub.py <fixture.npz> <cfg dir> <out dir> [yolo|rektnet]"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

fixture, cfg_dir, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
which = sys.argv[4] if len(sys.argv) > 4 else "yolo"
z = np.load(fixture)
cuda = torch.cuda.is_available()
device = torch.device("cuda:0" if cuda else "cpu")
rank = int(os.environ.get("RANK", "0"))

if which == "yolo":
    from models import Darknet                                                       # train.py:19 (resolves to dropin/CVC-YOLOv3/models.py)
    os.chdir(cfg_dir)
    model = Darknet(config_path="mini.cfg", xy_loss=2.0, wh_loss=1.6, no_object_loss=25.0, object_loss=0.1, vanilla_anchor=False)
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, weight_decay=0.0)
    model.load_weights("mini.weights", model.get_start_weight_dim())
    ndev = torch.cuda.device_count()
    if torch.cuda.device_count() > 1:
        print("Using ", torch.cuda.device_count(), " GPUs")
        model = nn.DataParallel(model)
    model = model.to(device, non_blocking=True)
    model.train()
    imgs, targets = torch.from_numpy(z["x"]), torch.from_numpy(z["targets"])          # the WHOLE batch, as the script's DataLoader delivers it
    imgs = imgs.to(device, non_blocking=True)
    targets = targets.to(device, non_blocking=True)
    targets.requires_grad_(False)
    optimizer.zero_grad()
    losses = model(imgs, targets)
    losses[0].sum().backward()
    grads = [p.grad.detach().clone() for p in model.parameters()]                    # (read before the update only because Adam rewrites nothing in .grad)
    optimizer.step()
    logged = [loss.sum().to("cpu").item() for loss in losses]
    first = losses[0].item()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.array(logged, np.float32), first=first, ndev=ndev,
             g0=grads[0].cpu().numpy(), glast=grads[-2].cpu().numpy(), gnorm=np.array([float(g.double().norm()) for g in grads]),
             wrapped=int(isinstance(model, nn.DataParallel)))
else:
    from keypoint_net import KeypointNet                                             # RektNet/train_eval.py:24-25
    from cross_ratio_loss import CrossRatioLoss
    zs = np.load(os.path.join(os.path.dirname(fixture), "rektnet_net.npz"))
    model = KeypointNet(7, (80, 80))
    model.load_state_dict({k[4:]: torch.from_numpy(zs[k]) for k in zs.files if k.startswith("sd::")})
    model = model.to(device)
    loss_function = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.train()
    x_batch = torch.from_numpy(z["x"]).to(device)
    y_points_batch = torch.from_numpy(z["tpts"]).to(device)
    y_hm_batch = torch.zeros(x_batch.shape[0], 7, 80, 80).to(device)
    optimizer.zero_grad()
    output = model(x_batch)
    loc_loss, geo_loss, loss = loss_function(output[0], output[1], y_hm_batch, y_points_batch)
    loss.backward()
    grads = [p.grad.detach().clone() for p in model.parameters()]
    optimizer.step()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.array([loc_loss.item(), geo_loss.item(), loss.item()], np.float32),
             ndev=torch.cuda.device_count(), g0=grads[0].cpu().numpy(), gnorm=np.array([float(g.double().norm()) for g in grads]))
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
