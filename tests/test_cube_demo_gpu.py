"""the reference's cube-localisation task (DatasetGradCAM: a cube of ones at a grid-aligned position, label = the
position index) as an end-to-end, self-checking run of the hot path: NeuroEncoder built with DATASET_NAME = 'gradcam'
(num_classes = (S / cube)^3, NeuroEncoder.py:179), the fused train step until the task is learned, then get_attention_map
(device-side Grad-CAM reduction): the map must concentrate on the cube.  The reference has no such test."""
import pytest
import torch

import weights as W

pytestmark = pytest.mark.gpu


def test_cube_localisation_train_then_gradcam_lights_up_the_cube():
    from neurovit_amd._cabi import require_gpu
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    from neurovit_amd.synthetic import cam_mass_in_cube, cube_volumes
    from neurovit_amd.trainer import TrainStep
    require_gpu()
    # patch 8 against cube 12: the patches at 8..16 straddle the cube faces.  (With faces ON patch borders every patch is constant and
    # the patch LayerNorm of vit_3d.py:93 maps them all to its bias: the task would be invisible to this architecture.)
    S, cube, patch = 40, 20, 8
    size = dict(TRAINING_VIT_DIM=128, TRAINING_VIT_DEPTH=2, TRAINING_VIT_HEADS=2, TRAINING_VIT_MLP_DIM=256)
    cfg = W.neuro_config(S, patch, dataset="gradcam", DEVICE="cuda", GRADCAM_CUBE_SIZE=cube, GRADCAM_THRESHOLD=25,
                         TRAINING_LEARNING_RATE=1e-3, TRAINING_WEIGHT_DECAY=1e-2, **size)
    torch.manual_seed(0)
    model = NeuroEncoder(cfg)
    assert model.volume_encoder.vit3d.mlp_head[1].out_features == (S // cube) ** 3      # NeuroEncoder.py:179
    vols, labels, corners = cube_volumes(160, S, cube, grid_noise=0.0, seed=1)
    tr_x, tr_y = vols[:128].cuda(), labels[:128].cuda()
    va_x, va_y, va_c = vols[128:].cuda(), labels[128:].cuda(), corners[128:]
    step = TrainStep(model)
    model.train()
    first = last = None
    for epoch in range(12):
        for i in range(0, 128, 16):
            loss = step(tr_x[i:i + 16], tr_y[i:i + 16])
            first = float(loss) if first is None else first
            last = float(loss)
    assert last < 0.2 * first, (first, last)
    model.eval()
    with torch.no_grad():
        acc = float((model(va_x).argmax(dim=1) == va_y).float().mean())
    assert acc >= 0.9, acc
    # Grad-CAM of the predicted class: the thresholded map must put far more of its mass into the cube than the 1/8 a uniform
    # map would, for most validation volumes
    model.train()                      # gradients of the encoder are needed (dropout 0 in this config)
    fractions = []
    for j in range(16):
        cam, cls = model.get_attention_map(va_x[j:j + 1])
        assert cam.shape == (S, S, S) and int(cls) == int(va_y[j])
        fractions.append(cam_mass_in_cube(cam, va_c[j], cube, margin=4))      # cube rounded out to patch borders: 24^3 of 40^3 = 0.216 by chance
    fractions.sort()
    print('cube demo: loss', first, '->', last, 'val acc', acc, 'cam mass in cube', fractions)
    assert fractions[len(fractions) // 2] > 0.5, fractions          # median; chance = 0.216
