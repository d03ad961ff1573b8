from .accuracy_calculator import CustomCalculator, get_accuracy_calculator
from .evaluate import evaluate, evaluate_multi_k, evaluate_sharded, get_tester
from .get_knn import get_knn
from .train_step import GradientAverager, backward_step, make_averager, train_step
from . import hamming

__all__ = ["CustomCalculator", "get_accuracy_calculator", "evaluate", "evaluate_multi_k", "evaluate_sharded", "get_tester",
           "get_knn", "hamming", "GradientAverager", "backward_step", "make_averager", "train_step"]
