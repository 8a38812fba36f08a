from .mlm_itm_head import MLMHead
from .ssl_head import NCEHeadForMM, NCEHeadForText, NCEHeadForVision

__all__ = ['NCEHeadForMM', 'NCEHeadForText', 'NCEHeadForVision', 'MLMHead']
