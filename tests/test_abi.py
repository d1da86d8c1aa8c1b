"""CPU: the C-ABI shared library loads and exports exactly what include/detbox.h declares
(no compute calls: there is no GPU here), and the host-side tables match the reference."""
import ctypes
import os
import re

import pytest
import torch

from helpers import manifest, sha

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'detbox.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(detbox_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    from object_detection_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from object_detection_amd import _lib, build
    build.build()                                  # hipcc cross-compiles for gfx950 without a GPU
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), name
    L = _lib.lib()
    assert L.detbox_version() >= 100
    # argument validation happens before any device work, so it is testable here
    assert L.detbox_box_convert(None, None, -1, 0, 0, None) == -1
    assert b'negative' in L.detbox_last_error()
    ws_sel = L.detbox_select_decode_workspace_bytes(16, 120087, 81, 1, 1000, 1)
    assert ws_sel > 16 * 80 * 1000 * 8                                  # the staging lists
    # no copy of the activated scores (4*B*P*C bytes) is reserved for the exact path, fused activation or not
    assert ws_sel < 4 * 16 * 120087 * 81 // 4 and ws_sel == L.detbox_select_decode_workspace_bytes(16, 120087, 81, 1, 1000, 0)
    assert L.detbox_select_decode_workspace_bytes(512, 120087, 81, 1, 1000, 1) < 4 * 512 * 120087 * 81 // 4
    # one buffer of kept runs (the tile merge reads it once); shapes beyond the tile merge keep the second one
    assert 2 * 4000 * 8 <= L.detbox_batched_nms_segments_workspace_bytes(2, 4000, 20) < 2 * 2 * 4000 * 8
    assert L.detbox_batched_nms_segments_workspace_bytes(2, 40000, 1000) >= 2 * 2 * 40000 * 8
    assert L.detbox_nms_rows_workspace_bytes(2, 4000, 20) == L.detbox_batched_nms_segments_workspace_bytes(2, 4000, 20)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from object_detection_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.DetboxError):
        _lib.lib()


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU error path')
def test_ops_refuse_to_run_without_gpu():
    import object_detection_amd as d
    with pytest.raises(d.ops._lib.DetboxError):
        d.box_iou(torch.zeros(2, 4), torch.zeros(3, 4))


def test_anchor_tables_bit_identical_to_reference():
    from object_detection_amd import anchors as A
    m = manifest()['anchor_tables']
    fk512, fk800, fksm = [64, 32, 16, 8, 4], [100, 50, 25, 13, 7], [8, 4, 2, 1, 1]
    T = {'ssd300_dboxes': A.ssd300_default_boxes(), 'ssd512_dboxes': A.ssd512_default_boxes(),
         'retinanet512_pboxes': A.retinanet_prior_boxes(fk512), 'retinanet800_pboxes': A.retinanet_prior_boxes(fk800),
         'retinanet_small_pboxes': A.retinanet_prior_boxes(fksm), 'fcos512_points': A.fcos_points(fk512),
         'fcos512_ranges': A.fcos_regress_ranges(fk512), 'fcos800_points': A.fcos_points(fk800),
         'fcos800_ranges': A.fcos_regress_ranges(fk800), 'yolov3_pboxes': A.yolov3_prior_boxes()}
    for k, v in T.items():
        assert list(v.shape) == m[k]['shape'] and sha(v) == m[k]['sha256'], k


def test_model_classes_mirror_reference_interface():
    import inspect
    import object_detection_amd as d
    for cls in (d.SSD300, d.SSD512, d.RetinaNet, d.FCOS, d.YoloV3):
        sig = inspect.signature(cls.pre_predict)
        assert list(sig.parameters)[:4] == ['self', 'outputs', 'conf_thresh', 'top_k']
        assert sig.parameters['conf_thresh'].default == 0.01 and sig.parameters['top_k'].default == 200
        assert list(inspect.signature(cls.loss).parameters)[:4] == ['self', 'outputs', 'gt_bboxes', 'gt_labels']
    assert inspect.signature(d.SSD300.loss).parameters['iou_thresh'].default == 0.5
    assert inspect.signature(d.RetinaNet.loss).parameters['iou_threshs'].default == (0.4, 0.5)
    p = inspect.signature(d.Predictor.__init__).parameters
    assert p['iou_thresh'].default == 0.45 and 'classes' in p and 'out_dir' in p
    assert d.SSD300(20).dboxes.shape == (8732, 4) and d.SSD512(20).dboxes.shape == (24564, 4)
    assert d.YoloV3(20).pboxes.shape == (10647, 5) and d.FCOS(20).points.shape == (5456, 2)


def test_release_workspaces_empties_the_cache():
    from object_detection_amd import ops
    ops._workspaces[(0, 'probe', 0)] = torch.empty(1024, dtype=torch.uint8)
    ops._workspaces[(1, 'probe', 0)] = torch.empty(512, dtype=torch.uint8)
    assert ops.release_workspaces('cuda:1') == 512 and (0, 'probe', 0) in ops._workspaces
    assert ops.release_workspaces() == 1024 and not ops._workspaces
