"""image-classification_amd -- MI355X-native feature-bank similarity path.

One hot path of sailist/image-classification, rebuilt for gfx950: the weighted-kNN monitor
(``models/knn.py``) and the MoCo queue (``models/memory_bank.py``, ``track_ssl/moco.py``).
Import as ``image_classification_amd`` (alias module at the repository root)."""
from . import _lib
from . import module_utils
from . import sharded
from .knn import BankBuilder, BankIndex, knn_merge, knn_predict, knn_predict_full, knn_topk, knn_topk_static, knn_vote, pack_rows
from .memory_bank import MemoryBank, StorageBank
from .moco import contrastive_loss2, moco_queue_loss
from .soft_knn import smooth_prob
from .neighbors import kth_largest, neighbor_label_agreement

__all__ = ["knn_predict", "knn_predict_full", "knn_topk", "knn_topk_static", "knn_vote", "knn_merge", "BankIndex", "BankBuilder",
           "pack_rows", "MemoryBank", "StorageBank", "contrastive_loss2", "moco_queue_loss", "smooth_prob", "kth_largest", "neighbor_label_agreement", "sharded", "module_utils", "_lib"]
