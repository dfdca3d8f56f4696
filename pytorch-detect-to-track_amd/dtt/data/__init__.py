"""Host-side data layer of the D&T path: ImageNet VID / DET image databases, frame-pair
roidb construction, aspect-ratio grouping, crop / pad batching and the VOC-style AP evaluation -- the reference's
lib/roi_data_layer/ and the lib/datasets/ pieces its D&T drivers use, restated for Python 3 on numpy / PIL / torch."""
from .factory import get_imdb  # noqa: F401
from .loader import roibatchLoader, sampler  # noqa: F401
from .roidb import combined_roidb  # noqa: F401
