"""Classification kriging: a scikit-learn classifier + kriging of the ilr-transformed probability residuals, one kriging
model per ilr coordinate (reference ck.py:15-291)."""
import numpy as np

from .compat import Krige, check_sklearn_model, validate_sklearn

validate_sklearn()

from scipy.linalg import helmert  # noqa: E402
from sklearn.metrics import accuracy_score  # noqa: E402
from sklearn.preprocessing import OneHotEncoder  # noqa: E402
from sklearn.svm import SVC  # noqa: E402


def closure(data, k=1.0):
    """Rescale every row so that it sums to k (ck.py:215-240)."""
    data = np.asarray(data, dtype=float)
    return k * data / data.sum(axis=1, keepdims=True)


def ilr_transformation(data):
    """Isometric log-ratio coordinates of compositions (rows; zeros are lifted to machine eps) on the Helmert basis
    (ck.py:243-266): (n, D) -> (n, D-1)."""
    logs = np.log(np.maximum(data, np.finfo(float).eps))
    return logs @ (-helmert(logs.shape[1]).T)


def inverse_ilr_transformation(data):
    """Back from ilr coordinates to closed compositions (ck.py:269-291): (n, D-1) -> (n, D)."""
    data = np.asarray(data, dtype=float)
    return closure(np.exp(data @ (-helmert(data.shape[1] + 1))))


class ClassificationKriging:
    """`classification_model` gives class probabilities from the predictors p; the residual between the one-hot truth and
    those probabilities is kriged in ilr space (n_classes - 1 `Krige` models, keyword arguments as compat.Krige)."""

    def __init__(self, classification_model=SVC(), method="ordinary", variogram_model="linear", n_closest_points=10, **krige_kw):
        check_sklearn_model(classification_model, task="classification")
        self.classification_model = classification_model
        self.n_closest_points = n_closest_points
        self._kriging_kwargs = dict(method=method, variogram_model=variogram_model, n_closest_points=n_closest_points, **krige_kw)
        Krige(**self._kriging_kwargs)  # validate the arguments now, as the reference constructor does

    def fit(self, p, x, y):
        self.classification_model.fit(p, y.ravel())
        print("Finished learning classification model")
        self.classes_ = self.classification_model.classes_
        ncoord = len(self.classes_) - 1
        proba_ilr = ilr_transformation(self.classification_model.predict_proba(p))
        self.onehotencode = OneHotEncoder(categories=[self.classes_])
        truth_ilr = ilr_transformation(np.asarray(self.onehotencode.fit_transform(y).todense()))
        self.krige = [Krige(**self._kriging_kwargs) for _ in range(ncoord)]
        for i, k in enumerate(self.krige):
            k.fit(x=x, y=truth_ilr[:, i] - proba_ilr[:, i])
        print("Finished kriging residuals")

    def krige_residual(self, x, **kwargs):
        return np.vstack([k.predict(x=x, **kwargs) for k in self.krige]).T

    def predict(self, p, x, **kwargs):
        proba_ilr = ilr_transformation(self.classification_model.predict_proba(p))
        proba = inverse_ilr_transformation(self.krige_residual(x, **kwargs) + proba_ilr)
        return np.argmax(proba, axis=1)

    def score(self, p, x, y, sample_weight=None, **kwargs):
        return accuracy_score(y_pred=self.predict(p, x, **kwargs), y_true=y, sample_weight=sample_weight)
