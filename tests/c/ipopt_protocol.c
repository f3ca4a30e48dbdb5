/* Test harness: drives the library's IPOPT-typed callbacks the way IPOPT's C interface does -- through function
 * pointers of IpStdCInterface.h's types (restated here; coin-or/Ipopt is not in the image), in IPOPT's call order:
 * structure queries with values == NULL, then eval_f(new_x = 1), eval_grad_f(0), eval_g(0), eval_jac_g(0),
 * eval_h(0, new_lambda = 1) at every point.  Compiled by tests/test_ipopt_adapters.py with gcc; it links against
 * nothing: the callbacks arrive as pointers. */
typedef double Number;
typedef int Index;
typedef int Bool;
typedef void* UserDataPtr;
typedef Bool (*Eval_F_CB)(Index n, Number* x, Bool new_x, Number* obj_value, UserDataPtr user_data);
typedef Bool (*Eval_Grad_F_CB)(Index n, Number* x, Bool new_x, Number* grad_f, UserDataPtr user_data);
typedef Bool (*Eval_G_CB)(Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr user_data);
typedef Bool (*Eval_Jac_G_CB)(Index n, Number* x, Bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol,
                              Number* values, UserDataPtr user_data);
typedef Bool (*Eval_H_CB)(Index n, Number* x, Bool new_x, Number obj_factor, Index m, Number* lambda, Bool new_lambda,
                          Index nele_hess, Index* iRow, Index* jCol, Number* values, UserDataPtr user_data);

/* the five callbacks as CreateIpoptProblem receives them */
typedef struct {
  Eval_F_CB eval_f;
  Eval_G_CB eval_g;
  Eval_Grad_F_CB eval_grad_f;
  Eval_Jac_G_CB eval_jac_g;
  Eval_H_CB eval_h;
} callbacks;

int drive_structure(const callbacks* cb, Index n, Index m, Index nele_jac, Index nele_hess, Index* jRow, Index* jCol,
                    Index* hRow, Index* hCol, UserDataPtr ud) {
  if (!cb->eval_jac_g(n, 0, 0, m, nele_jac, jRow, jCol, 0, ud)) return 0;
  if (!cb->eval_h(n, 0, 0, 1.0, m, 0, 0, nele_hess, hRow, hCol, 0, ud)) return 0;
  return 1;
}

/* one IPOPT iteration's evaluations at x (scaled space), multipliers lambda, objective factor sigma */
int drive_point(const callbacks* cb, Index n, Index m, Index nele_jac, Index nele_hess, Number* x, Number sigma,
                Number* lambda, Number* f, Number* grad, Number* g, Number* jac, Number* hess, UserDataPtr ud) {
  if (!cb->eval_f(n, x, 1, f, ud)) return 0;
  if (!cb->eval_grad_f(n, x, 0, grad, ud)) return 0;
  if (!cb->eval_g(n, x, 0, m, g, ud)) return 0;
  if (!cb->eval_jac_g(n, x, 0, m, nele_jac, 0, 0, jac, ud)) return 0;
  if (!cb->eval_h(n, x, 0, sigma, m, lambda, 1, nele_hess, 0, 0, hess, ud)) return 0;
  return 1;
}

/* a wrong size must be refused, not evaluated */
int drive_wrong_sizes(const callbacks* cb, Index n, Index m, Index nele_jac, Number* x, Number* g, UserDataPtr ud) {
  Number f;
  return cb->eval_f(n + 1, x, 1, &f, ud) == 0 && cb->eval_g(n, x, 1, m + 1, g, ud) == 0 &&
         cb->eval_jac_g(n, x, 1, m, nele_jac + 1, 0, 0, g, ud) == 0;
}
