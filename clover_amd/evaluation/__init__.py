from .retrieval import (evaluate_retrieval, multi_gpu_test_retrieval, normalize_fn,
                        recall_for_video_text_retrieval)

__all__ = ['normalize_fn', 'recall_for_video_text_retrieval', 'multi_gpu_test_retrieval', 'evaluate_retrieval']
