"""Regression kriging: a scikit-learn regressor on the predictors + kriging of its residuals (reference rk.py:11-186)."""
from .compat import Krige, check_sklearn_model, validate_sklearn

validate_sklearn()

from sklearn.metrics import r2_score  # noqa: E402
from sklearn.svm import SVR  # noqa: E402


class RegressionKriging:
    """`regression_model` learns y from the predictors p; a `Krige` (all other keyword arguments, see compat.Krige) kriges
    y - model(p) at the coordinates x; predict() adds the two (rk.py:106-166)."""

    def __init__(self, regression_model=SVR(), method="ordinary", variogram_model="linear", n_closest_points=10, **krige_kw):
        check_sklearn_model(regression_model)
        self.regression_model = regression_model
        self.n_closest_points = n_closest_points
        self.krige = Krige(method=method, variogram_model=variogram_model, n_closest_points=n_closest_points, **krige_kw)

    def fit(self, p, x, y):
        self.regression_model.fit(p, y)
        print("Finished learning regression model")
        self.krige.fit(x=x, y=y - self.regression_model.predict(p))
        print("Finished kriging residuals")

    def krige_residual(self, x, **kwargs):
        return self.krige.predict(x, **kwargs)

    def predict(self, p, x, **kwargs):
        return self.krige_residual(x, **kwargs) + self.regression_model.predict(p)

    def score(self, p, x, y, sample_weight=None, **kwargs):
        return r2_score(y_pred=self.predict(p, x, **kwargs), y_true=y, sample_weight=sample_weight)
