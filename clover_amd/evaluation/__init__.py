from .retrieval import normalize_fn, recall_for_video_text_retrieval

__all__ = ['normalize_fn', 'recall_for_video_text_retrieval']
