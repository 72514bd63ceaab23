"""Differentiable evaluation of the right-hand side.  Not built yet:
the forward (inference) path is native; asking for gradients fails loudly rather than silently
falling back to a PyTorch composite."""


def rhs_with_grad(func, x):
  raise NotImplementedError(
    '%s.forward was asked for gradients: the native VJP of the fused right-hand side is the next row of '
    'SURVEY.md section 8f; run under torch.no_grad() / model.eval() for the forward solve' % func.__class__.__name__)
