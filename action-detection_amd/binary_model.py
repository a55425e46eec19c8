"""Mirror of the reference's ``binary_model.BinaryClassifier`` (/root/reference/binary_model.py:7-300), the
actionness classifier the TAG proposal pipeline trains on the same backbone (``binary_train.py`` /
``binary_test.py``), on the MI355X kernels: backbone -> (Dropout) -> mean over the ``course_segment`` snippets
-> ``classifier_fc``; test mode scores single frames with ``test_fc``.

Same constructor, ``forward(inputdata, target)`` -> ``(logits, target)``, ``prepare_test_fc``, ``train()`` BN
freezing and ``get_optim_policies`` as the reference.  Backbones: BNInception and InceptionV3, as for ``SSN``.  The segment mean runs on the STPP kernels (one part = all segments).
"""
import torch
from torch import nn

from . import functional as FN
from . import kernels as K
from .ops.ssn_ops import Identity
from .ssn_models import SSN, HipDropout, HipLinear


class BinaryClassifier(torch.nn.Module):
    def __init__(self, num_class, course_segment, modality,
                 base_model='resnet101', new_length=None,
                 dropout=0.8,
                 crop_num=1, test_mode=False, bn_mode='frozen', verbose=False):
        super(BinaryClassifier, self).__init__()
        self.modality = modality
        self.num_segments = course_segment
        self.course_segment = course_segment
        self.reshape = True
        self.dropout = dropout
        self.crop_num = crop_num
        self.test_mode = test_mode
        self.bn_mode = bn_mode
        if new_length is None:
            self.new_length = 1 if modality == "RGB" else 5
        else:
            self.new_length = new_length
        if verbose:
            print("Initializing BinaryClassifier with base model: {} ({} / {} segments / dropout {} / bn_mode {})".format(
                base_model, modality, course_segment, dropout, bn_mode))

        # backbone choice, input statistics and the first-conv surgery are the ones of SSN (ssn_models.py:107-154,
        # 318-343 == binary_model.py:154-199, 54-79)
        SSN._prepare_base_model(self, base_model)
        self._prepare_binary_classifier(num_class)
        if self.modality == 'Flow':
            self.base_model = SSN._construct_flow_model(self, self.base_model)
        elif self.modality == 'RGBDiff':
            raise NotImplementedError("RGBDiff modality is outside the built hot path (SURVEY.md section 8f-4)")
        self.prepare_bn()
        # mean over the course segments = the activity feature of an STPP whose only part spans all segments
        self._table = K.make_stpp_table([(0, course_segment, 1, -1)], course_segment, 0, course_segment)

    # ---- /root/reference/binary_model.py:120-136
    def _prepare_binary_classifier(self, num_class):
        feature_dim = getattr(self.base_model, self.base_model.last_layer_name).in_features
        if self.dropout == 0:
            setattr(self.base_model, self.base_model.last_layer_name, Identity())
        else:
            setattr(self.base_model, self.base_model.last_layer_name, HipDropout(p=self.dropout))
        self.classifier_fc = HipLinear(feature_dim, num_class)
        nn.init.normal_(self.classifier_fc.weight.data, 0, 0.001)
        nn.init.constant_(self.classifier_fc.bias.data, 0)
        self.test_fc = None
        self.feature_dim = feature_dim
        return feature_dim

    prepare_bn = SSN.prepare_bn                 # binary_model.py:139-150 == ssn_models.py:95-105
    get_optim_policies = SSN.get_optim_policies     # binary_model.py:259-300 == ssn_models.py:203-251
    _backbone = SSN._backbone
    crop_size = SSN.crop_size
    scale_size = SSN.scale_size

    # ---- /root/reference/binary_model.py:203-216
    def train(self, mode=True):
        super(BinaryClassifier, self).train(mode)
        count = 0
        if self.freeze_count is None:
            return self
        for m in self.base_model.modules():
            if isinstance(m, nn.BatchNorm2d):
                count += 1
                if count >= self.freeze_count:
                    m.eval()
                    m.weight.requires_grad = False
                    m.bias.requires_grad = False
        return self

    # ---- /root/reference/binary_model.py:219-240
    def forward(self, inputdata, target=None):
        if not self.test_mode:
            return self.train_forward(inputdata, target)
        return self.test_forward(inputdata)

    def train_forward(self, inputdata, target):
        base_out = self._backbone(inputdata)
        n = base_out.shape[0] // self.course_segment
        ones = torch.ones((n, 2), device=base_out.device, dtype=torch.float32)
        course_ft, _ = FN.StppFn.apply(base_out, ones, self._table, self.course_segment)
        raw_course_ft = self.classifier_fc(course_ft)
        return raw_course_ft, target.reshape(-1).to(raw_course_ft.device)

    def test_forward(self, input):
        base_out = self._backbone(input)
        return self.test_fc(base_out), base_out

    # ---- /root/reference/binary_model.py:245-254
    def prepare_test_fc(self):
        self.test_fc = HipLinear(self.classifier_fc.in_features, self.classifier_fc.out_features)
        self.test_fc.weight.data = self.classifier_fc.weight.data
        self.test_fc.bias.data = self.classifier_fc.bias.data
